// resample.hip -- AudioResampler / BatchResampler behind the C ABI.
//
// Reference surface: src-tauri/src/modules/audio/resampler.rs
//   AudioResampler::{new :32, create_48k_to_16k :60, process :71, getters :96-108}
//   BatchResampler::{new :121, process :132, flush :150}
// Equal rates are an exact passthrough (:72-75, test :185-190) [R].
// Unequal rates: the reference forwards to rubato's cubic FastFixedIn, which no
// reference test ever executes; the arithmetic here is
// the polyphase windowed-sinc resampler of docs/OPS.md [B], the algorithm the
// reference's own spec asks for (specs/0001-spec.md:266,
// specs/0002-design.md:1519-1523: 64 taps per phase, Blackman-Harris).
//
//   y[m] = sum_{k<T} h[(m*M mod L) + k*L] * x[floor(m*M/L) - k + T/2]
//
// Device path: ratios whose 16-output blocks repeat with a short period (every
// ratio on BASELINE's list) run as a banded matrix product on MFMA (banded.hpp).
// Any other ratio runs the VALU kernel below: prototype phase-major ([L][T], rows
// padded to T+4 floats) in LDS, the input span of a 1024-output tile staged in LDS
// with coalesced loads, one thread per output.
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <vector>

#include "af_common.hpp"
#include "banded.hpp"

namespace af {
namespace rs {

constexpr int kT = banded::kResamplerTaps;   // taps per phase
constexpr int kRow = kT + 4;            // padded LDS row
constexpr int kBlock = 256;
constexpr int kOutPerThread = 4;
constexpr int kTileOut = kBlock * kOutPerThread;   // outputs per workgroup tile
constexpr double kPi = 3.14159265358979323846;

// Device staging of the host-buffer convenience entry point af_resample_batch, one slot per device ordinal.
constexpr size_t kKeepBytes = (size_t)256 << 20;        // what a slot may keep between calls
struct Staging {
    std::mutex mu;                                       // one caller at a time per device
    DevBuf in, out;
};
inline Staging& staging_slot(int dev)
{
    static Staging* const slots = new Staging[kMaxDevices];   // never freed: no hipFree during static destruction
    return slots[dev >= 0 && dev < kMaxDevices ? dev : 0];
}

// Cubic mode: the polynomial through (-1, y0), (0, y1), (1, y2), (2, y3) evaluated at t in [0, 1) (Horner, f32);
// the position m * in / out and its split into sample index and fraction are float64.
__global__ void cubic_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n_in, int64_t n_out,
                             int64_t total, double step)
{
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t clip = idx / n_out, m = idx - clip * n_out;
        const double pos = (double)m * step;
        const int64_t i = (int64_t)pos;                       // pos >= 0: truncation = floor
        const float t = (float)(pos - (double)i);
        const float* xc = x + clip * n_in;
        const float y0 = i - 1 >= 0 && i - 1 < n_in ? xc[i - 1] : 0.0f;
        const float y1 = i < n_in ? xc[i] : 0.0f;
        const float y2 = i + 1 < n_in ? xc[i + 1] : 0.0f;
        const float y3 = i + 2 < n_in ? xc[i + 2] : 0.0f;
        const float a1 = -y0 * (1.0f / 3.0f) - y1 * 0.5f + y2 - y3 * (1.0f / 6.0f);
        const float a2 = (y0 + y2) * 0.5f - y1;
        const float a3 = (y1 - y2) * 0.5f + (y3 - y0) * (1.0f / 6.0f);
        y[idx] = y1 + t * (a1 + t * (a2 + t * a3));
    }
}

struct Geometry {
    const float* x;          // clip 0, element at absolute stream index x_abs0
    float* y;
    const float* taps;       // [L][kRow] phase-major (padded rows)
    int64_t x_abs0;          // absolute index of x[0]
    int64_t x_len;           // valid samples per clip in the buffer
    int64_t x_stride;        // elements between clips
    int64_t y_stride;
    int64_t m0;              // absolute index of the first output to compute
    int64_t n_out;           // outputs per clip
    int64_t tiles_per_clip;
    int64_t n_tiles;         // tiles_per_clip * n_clips
    uint32_t L, M;
    int taps_in_lds;
    int x_tile_cap;          // floats reserved for the input span
};

__global__ __launch_bounds__(kBlock) void resample_kernel(const Geometry g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                                   // input span
    float* hs = smem + g.x_tile_cap;                    // taps (optional)
    const int tid = threadIdx.x;

    if (g.taps_in_lds) {
        const int n4 = (int)(g.L * kRow) / 4;
        for (int i = tid; i < n4; i += kBlock)
            reinterpret_cast<float4*>(hs)[i] = reinterpret_cast<const float4*>(g.taps)[i];
    }
    const float* taps = g.taps_in_lds ? hs : g.taps;

    for (int64_t tile = blockIdx.x; tile < g.n_tiles; tile += gridDim.x) {
        const int64_t clip = tile / g.tiles_per_clip;
        const int64_t t_in_clip = tile - clip * g.tiles_per_clip;
        const int64_t mt0 = g.m0 + t_in_clip * kTileOut;             // absolute output index
        int64_t cnt = g.n_out - t_in_clip * kTileOut;
        if (cnt > kTileOut) cnt = kTileOut;
        // absolute input span [a0, a1) needed by outputs [mt0, mt0+cnt)
        const int64_t q_lo = (mt0 * (int64_t)g.M) / g.L;
        const int64_t q_hi = ((mt0 + cnt - 1) * (int64_t)g.M) / g.L;
        const int64_t a0 = q_lo - kT / 2 + 1;
        const int64_t a1 = q_hi + kT / 2 + 1;
        const float* xc = g.x + clip * g.x_stride;
        __syncthreads();                                 // previous tile's readers done
        for (int64_t i = tid; i < a1 - a0; i += kBlock) {
            const int64_t rel = a0 + i - g.x_abs0;       // index into the clip buffer
            xs[i] = (rel >= 0 && rel < g.x_len) ? xc[rel] : 0.0f;   // zero extension
        }
        __syncthreads();
        float* yc = g.y + clip * g.y_stride + (mt0 - g.m0);
#pragma unroll
        for (int j = 0; j < kOutPerThread; ++j) {
            const int o = tid + j * kBlock;
            if (o < cnt) {
                const int64_t mm = (mt0 + o) * (int64_t)g.M;
                const int64_t q = mm / g.L;
                const uint32_t r = (uint32_t)(mm - q * g.L);
                const float* h = taps + (size_t)r * kRow;
                const float* xp = xs + (q + kT / 2 - a0);            // x[q + T/2], k = 0
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < kT; k += 4) {
                    const float4 hv = *reinterpret_cast<const float4*>(h + k);
                    acc += hv.x * xp[-k];
                    acc += hv.y * xp[-k - 1];
                    acc += hv.z * xp[-k - 2];
                    acc += hv.w * xp[-k - 3];
                }
                yc[o] = acc;
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------
struct Design {
    uint32_t in_hz = 0, out_hz = 0, L = 1, M = 1;
    std::vector<float> proto;        // h[L*T] in prototype order (f32)
    DevBuf d_taps;                   // [L][kRow]
    banded::Plan plan;               // MFMA path when plan.ok
    bool valu_ok = false;            // the thread-per-output kernel is set up as well
    banded::Plan plan2;              // same, interleaved stereo input downmixed while staged
    DevBuf d_afrag;
    int x_tile_cap = 0;
    size_t smem = 0;
    int taps_in_lds = 0;
};

// docs/OPS.md "Polyphase resampler": periodic Blackman-Harris windowed sinc of
// length L*T centred on L*T/2, cutoff 0.5/max(L,M) cycles per upsampled
// sample, each phase normalised to unit DC gain.
void design_prototype(uint32_t L, uint32_t M, std::vector<double>& h)
{
    const size_t LT = (size_t)L * kT;
    h.resize(LT);
    const double c = (double)(LT / 2);
    const double fcu = 0.5 / (double)(L > M ? L : M);
    const double a0 = 0.35875, a1 = 0.48829, a2 = 0.14128, a3 = 0.01168;
    for (size_t p = 0; p < LT; ++p) {
        const double t = (double)p - c;
        const double x = 2.0 * fcu * t;
        const double s = t == 0.0 ? 1.0 : std::sin(kPi * x) / (kPi * x);
        const double ph = 2.0 * kPi * (double)p / (double)LT;
        h[p] = s * (a0 - a1 * std::cos(ph) + a2 * std::cos(2.0 * ph) - a3 * std::cos(3.0 * ph));
    }
    for (uint32_t r = 0; r < L; ++r) {
        double sum = 0.0;
        for (int k = 0; k < kT; ++k) sum += h[r + (size_t)k * L];
        for (int k = 0; k < kT; ++k) h[r + (size_t)k * L] /= sum;
    }
}

af_status make_design(uint32_t in_hz, uint32_t out_hz, Design& d, bool upload)
{
    AF_REQUIRE(in_hz > 0 && out_hz > 0, "sample rates must be positive (got %u -> %u)", in_hz, out_hz);
    const uint32_t g = std::gcd(in_hz, out_hz);
    d.in_hz = in_hz;
    d.out_hz = out_hz;
    d.L = out_hz / g;
    d.M = in_hz / g;
    if ((uint64_t)d.L * kRow * sizeof(float) > (1ull << 30))
        return fail(AF_ERR_UNSUPPORTED, "ratio %u/%u needs too many phases", d.L, d.M);
    std::vector<double> h;
    design_prototype(d.L, d.M, h);
    d.proto.resize(h.size());
    for (size_t i = 0; i < h.size(); ++i) d.proto[i] = (float)h[i];
    d.plan = banded::make_plan(d.L, d.M, kT, kT / 2);
    d.plan2 = banded::make_plan(d.L, d.M, kT, kT / 2, 2);
    if (d.plan.ok) {
        if (upload) {
            DevBuf d_proto;
            AF_TRY(d_proto.reserve(d.proto.size() * sizeof(float)));
            AF_TRY(d.d_afrag.reserve(d.plan.afrag_floats() * sizeof(float)));
            AF_HIP(hipMemcpy(d_proto.p, d.proto.data(), d.proto.size() * sizeof(float), hipMemcpyHostToDevice));
            AF_TRY(banded::build_afrag(d.plan, d_proto.as<float>(), d.d_afrag.as<float>(), nullptr));
            AF_HIP(hipStreamSynchronize(nullptr));
        }
    }
    // The thread-per-output kernel's geometry as well: it serves the ratios the MFMA plan does not
    // cover and the clips beyond its 32-bit offsets.
    // input span of one tile: ceil(kTileOut*M/L) + T + 2
    d.x_tile_cap = (int)(((uint64_t)kTileOut * d.M + d.L - 1) / d.L) + kT + 4;
    d.x_tile_cap = (d.x_tile_cap + 3) & ~3;
    const size_t taps_bytes = (size_t)d.L * kRow * sizeof(float);
    d.taps_in_lds = (d.x_tile_cap * sizeof(float) + taps_bytes) <= 150 * 1024 ? 1 : 0;
    d.smem = d.x_tile_cap * sizeof(float) + (d.taps_in_lds ? taps_bytes : 0);
    d.valu_ok = d.x_tile_cap * sizeof(float) <= 150 * 1024;
    if (!d.valu_ok) {
        if (d.plan.ok) return AF_OK;
        return fail(AF_ERR_UNSUPPORTED, "downsampling ratio %u/%u too steep for one LDS tile", d.L, d.M);
    }
    if (upload) {
        std::vector<float> rows((size_t)d.L * kRow, 0.0f);
        for (uint32_t r = 0; r < d.L; ++r)
            for (int k = 0; k < kT; ++k) rows[(size_t)r * kRow + k] = d.proto[r + (size_t)k * d.L];
        AF_TRY(d.d_taps.reserve(rows.size() * sizeof(float)));
        AF_HIP(hipMemcpy(d.d_taps.p, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return AF_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

af_status launch(const Design& d, const float* x_dev, int64_t x_abs0, int64_t x_len, int64_t x_stride,
                 float* y_dev, int64_t y_stride, int64_t m0, int64_t n_out, int64_t n_clips,
                 hipStream_t stream)
{
    if (n_out <= 0 || n_clips <= 0) return AF_OK;
    if (banded::fits(d.plan, x_len) && !debug_knob(kKnobBandedOff))      // (the switch also works on a cached design)
        return banded::launch(d.plan, d.d_afrag.as<float>(), x_dev, x_abs0, x_len, x_stride, y_dev,
                              y_stride, m0, n_out, n_clips, stream);
    if (!d.valu_ok)
        return fail(AF_ERR_UNSUPPORTED, "clips of %lld samples at ratio %u/%u: beyond the MFMA kernel's 32-bit offsets "
                    "and too steep for the thread-per-output kernel", (long long)x_len, d.L, d.M);
    Geometry g{};
    g.x = x_dev;
    g.y = y_dev;
    g.taps = d.d_taps.as<float>();
    g.x_abs0 = x_abs0;
    g.x_len = x_len;
    g.x_stride = x_stride;
    g.y_stride = y_stride;
    g.m0 = m0;
    g.n_out = n_out;
    g.tiles_per_clip = ceil_div(n_out, kTileOut);
    g.n_tiles = g.tiles_per_clip * n_clips;
    g.L = d.L;
    g.M = d.M;
    g.taps_in_lds = d.taps_in_lds;
    g.x_tile_cap = d.x_tile_cap;
    static LdsAttrFlags attr_set;
    AF_HIP(allow_big_lds(resample_kernel, 160 * 1024, attr_set));
    int64_t grid = g.n_tiles;
    const int64_t cap = device_cu_count(current_device()) * (int64_t)(d.smem > 80 * 1024 ? 1 : d.smem > 53 * 1024 ? 2 : 3) * 4;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)grid), dim3(kBlock), d.smem, stream, g);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

}  // namespace rs
}  // namespace af

using namespace af;

// AudioResampler (resampler.rs:12-21): rates + the streaming state the inner
// rubato resampler would carry (here: stream position and T samples of history).
struct af_resampler {
    rs::Design d;
    bool passthrough = false;
    int64_t n_in_total = 0;      // samples consumed so far
    int64_t n_out_total = 0;     // outputs emitted so far
    std::vector<float> hist;     // last min(T, n_in_total) input samples
    DevBuf d_x, d_y;
};

namespace {

// number of outputs whose whole support lies in the first n input samples
int64_t outputs_ready(const af_resampler* r, int64_t n)
{
    if (n < rs::kT / 2) return 0;
    return rs::ceil_div((n - rs::kT / 2) * (int64_t)r->d.L, r->d.M);
}

af_status emit(af_resampler* r, const float* in, size_t n_in, int64_t m_end, float* out,
               size_t out_cap, size_t* n_out)
{
    const int64_t cnt = m_end - r->n_out_total;
    if (n_out) *n_out = cnt > 0 ? (size_t)cnt : 0;
    const int64_t h = (int64_t)r->hist.size();
    if (cnt > 0) {
        if ((size_t)cnt > out_cap)
            return fail(AF_ERR_BUFFER_TOO_SMALL, "resampler output needs %lld floats, have %zu",
                        (long long)cnt, out_cap);
        AF_TRY(ensure_device(-1));
        const size_t nx = (size_t)h + n_in;
        AF_TRY(r->d_x.reserve((nx ? nx : 1) * sizeof(float)));
        AF_TRY(r->d_y.reserve((size_t)cnt * sizeof(float)));
        if (h) AF_HIP(hipMemcpy(r->d_x.p, r->hist.data(), h * sizeof(float), hipMemcpyHostToDevice));
        if (n_in)
            AF_HIP(hipMemcpy(r->d_x.as<float>() + h, in, n_in * sizeof(float), hipMemcpyHostToDevice));
        AF_TRY(rs::launch(r->d, r->d_x.as<float>(), r->n_in_total - h, (int64_t)nx, 0,
                          r->d_y.as<float>(), 0, r->n_out_total, cnt, 1, nullptr));
        AF_HIP(hipMemcpy(out, r->d_y.p, (size_t)cnt * sizeof(float), hipMemcpyDeviceToHost));
        r->n_out_total = m_end;
    }
    // carry the last T samples of the stream
    std::vector<float> nh;
    const size_t keep = rs::kT;
    if (n_in >= keep) {
        nh.assign(in + (n_in - keep), in + n_in);
    } else {
        const size_t from_old = std::min((size_t)h, keep - n_in);
        nh.assign(r->hist.end() - from_old, r->hist.end());
        nh.insert(nh.end(), in, in + n_in);
    }
    r->hist.swap(nh);
    r->n_in_total += (int64_t)n_in;
    return AF_OK;
}

}  // namespace

extern "C" {

af_status af_resampler_create(uint32_t input_rate, uint32_t output_rate, af_resampler** out)
{
    AF_REQUIRE(out, "af_resampler_create: null output");
    *out = nullptr;
    AF_REQUIRE(input_rate > 0 && output_rate > 0, "sample rates must be positive (got %u -> %u)",
               input_rate, output_rate);
    af_resampler* r = new (std::nothrow) af_resampler();
    if (!r) return fail(AF_ERR_OUT_OF_MEMORY, "out of host memory");
    r->passthrough = input_rate == output_rate;          // resampler.rs:33-40
    r->d.in_hz = input_rate;
    r->d.out_hz = output_rate;
    if (!r->passthrough) {
        af_status st = ensure_device(-1);
        if (st == AF_OK) st = rs::make_design(input_rate, output_rate, r->d, true);
        if (st != AF_OK) { delete r; return st; }
    }
    *out = r;
    return AF_OK;
}

af_status af_resampler_create_48k_to_16k(af_resampler** out)
{   // resampler.rs:60-62
    return af_resampler_create(48000, 16000, out);
}

void af_resampler_destroy(af_resampler* r) { delete r; }
uint32_t af_resampler_input_rate(const af_resampler* r) { return r ? r->d.in_hz : 0; }
uint32_t af_resampler_output_rate(const af_resampler* r) { return r ? r->d.out_hz : 0; }
int af_resampler_needs_resampling(const af_resampler* r) { return r && r->d.in_hz != r->d.out_hz; }

size_t af_resampler_out_len(const af_resampler* r, size_t n_in)
{
    if (!r) return 0;
    if (r->passthrough) return n_in;
    // everything that could still be pending plus this call's share
    const int64_t total = r->n_in_total + (int64_t)n_in;
    const int64_t m_end = rs::ceil_div(total * (int64_t)r->d.L, r->d.M);
    const int64_t cnt = m_end - r->n_out_total;
    return cnt > 0 ? (size_t)cnt : 0;
}

af_status af_resampler_process(af_resampler* r, const float* in, size_t n_in, float* out,
                               size_t out_cap, size_t* n_out)
{
    AF_REQUIRE(r, "af_resampler_process: null handle");
    AF_REQUIRE(in || n_in == 0, "af_resampler_process: null input");
    if (r->passthrough) {                                 // resampler.rs:72-75
        if (n_out) *n_out = n_in;
        if (out_cap < n_in) return fail(AF_ERR_BUFFER_TOO_SMALL, "need %zu floats", n_in);
        if (n_in) std::memcpy(out, in, n_in * sizeof(float));
        return AF_OK;
    }
    const int64_t m_end = outputs_ready(r, r->n_in_total + (int64_t)n_in);
    return emit(r, in, n_in, m_end > r->n_out_total ? m_end : r->n_out_total, out, out_cap, n_out);
}

af_status af_resampler_flush(af_resampler* r, float* out, size_t out_cap, size_t* n_out)
{
    AF_REQUIRE(r, "af_resampler_flush: null handle");
    if (r->passthrough) { if (n_out) *n_out = 0; return AF_OK; }
    const int64_t m_end = rs::ceil_div(r->n_in_total * (int64_t)r->d.L, r->d.M);
    return emit(r, nullptr, 0, m_end > r->n_out_total ? m_end : r->n_out_total, out, out_cap, n_out);
}

void af_resampler_reset(af_resampler* r)
{
    if (!r) return;
    r->n_in_total = r->n_out_total = 0;
    r->hist.clear();
}

size_t af_resample_out_len(uint32_t input_rate, uint32_t output_rate, size_t n_in)
{
    if (!input_rate || !output_rate) return 0;
    const uint32_t g = std::gcd(input_rate, output_rate);
    const unsigned __int128 p = (unsigned __int128)n_in * (output_rate / g);
    const uint32_t M = input_rate / g;
    return (size_t)((p + M - 1) / M);
}

}  // extern "C"

namespace {

// one cached design per (in, out) pair for the stateless batch calls (process-wide, immutable
// once built; the device tables belong to the device that was current at first use)
af_status cached_design(uint32_t in_hz, uint32_t out_hz, rs::Design** out)
{
    // a design owns device tables: one per (ratio, device ordinal)
    static std::mutex mu;
    static std::vector<std::pair<int, rs::Design*>> cache;
    std::lock_guard<std::mutex> lock(mu);
    const int dev = current_device();
    for (const auto& e : cache)
        if (e.first == dev && e.second->in_hz == in_hz && e.second->out_hz == out_hz) { *out = e.second; return AF_OK; }
    rs::Design* d = new (std::nothrow) rs::Design();
    if (!d) return fail(AF_ERR_OUT_OF_MEMORY, "out of host memory");
    af_status st = rs::make_design(in_hz, out_hz, *d, true);
    if (st != AF_OK) { delete d; return st; }
    cache.push_back({dev, d});
    *out = d;
    return AF_OK;
}

}  // namespace

namespace af {

// The cached design's MFMA plan and fragments for the fused chain stage (chain.hip).
af_status resampler_fragments(uint32_t in_hz, uint32_t out_hz, int ch, banded::Plan* plan, const float** afrag_dev)
{
    rs::Design* d = nullptr;
    AF_TRY(cached_design(in_hz, out_hz, &d));
    *plan = ch == 2 ? d->plan2 : d->plan;
    *afrag_dev = d->d_afrag.as<float>();
    return AF_OK;
}

}  // namespace af

extern "C" {

af_status af_resample_batch_device(uint32_t input_rate, uint32_t output_rate, const float* in_dev,
                                   size_t n_clips, size_t n_in, float* out_dev, void* stream)
{
    AF_REQUIRE(input_rate > 0 && output_rate > 0, "sample rates must be positive");
    if (n_clips == 0 || n_in == 0) return AF_OK;
    AF_REQUIRE(in_dev && out_dev, "af_resample_batch_device: null buffer");
    if (input_rate == output_rate) {
        AF_HIP(hipMemcpyAsync(out_dev, in_dev, n_clips * n_in * sizeof(float), hipMemcpyDeviceToDevice,
                              as_stream(stream)));
        return AF_OK;
    }
    rs::Design* d = nullptr;
    AF_TRY(cached_design(input_rate, output_rate, &d));
    const int64_t n_out = (int64_t)af_resample_out_len(input_rate, output_rate, n_in);
    return rs::launch(*d, in_dev, 0, (int64_t)n_in, (int64_t)n_in, out_dev, n_out, 0, n_out,
                      (int64_t)n_clips, as_stream(stream));
}

int af_resample_plan_describe(uint32_t input_rate, uint32_t output_rate, uint32_t channels, int32_t* info)
{
    // host-only: which kernel shape a ratio gets (tests/test_host_mirror.py checks the invariants)
    if (!info || !input_rate || !output_rate || (channels != 1 && channels != 2)) return 0;
    const uint32_t g = std::gcd(input_rate, output_rate);
    const banded::Plan p = banded::make_plan(output_rate / g, input_rate / g, rs::kT, rs::kT / 2, (int)channels);
    const int32_t v[12] = {p.ok ? 1 : 0, (int32_t)p.L, (int32_t)p.M, p.nb, p.S, p.NK, p.G, p.wpp,
                           p.nwaves, p.seg_len, p.seg_stride, (int32_t)p.smem};
    for (int i = 0; i < 12; ++i) info[i] = v[i];
    return p.ok ? 1 : 0;
}

af_status af_resample_prepare(uint32_t input_rate, uint32_t output_rate)
{
    AF_REQUIRE(input_rate > 0 && output_rate > 0, "sample rates must be positive");
    if (input_rate == output_rate) return AF_OK;
    AF_TRY(ensure_device(-1));
    rs::Design* d = nullptr;
    return cached_design(input_rate, output_rate, &d);
}

af_status af_resample_batch_stereo_device(uint32_t input_rate, uint32_t output_rate,
                                          const float* in_dev, size_t n_clips, size_t n_frames,
                                          float* out_dev, void* stream)
{
    AF_REQUIRE(input_rate > 0 && output_rate > 0, "sample rates must be positive");
    if (n_clips == 0 || n_frames == 0) return AF_OK;
    AF_REQUIRE(in_dev && out_dev, "af_resample_batch_stereo_device: null buffer");
    if (input_rate == output_rate)
        return af_to_mono_device(in_dev, n_clips * n_frames * 2, 2, out_dev, stream);
    rs::Design* d = nullptr;
    AF_TRY(cached_design(input_rate, output_rate, &d));
    if (!d->plan2.ok)
        return fail(AF_ERR_UNSUPPORTED,
                    "%u -> %u Hz has no fused stereo path: call af_to_mono_device, then af_resample_batch_device",
                    input_rate, output_rate);
    const int64_t n_out = (int64_t)af_resample_out_len(input_rate, output_rate, n_frames);
    return banded::launch(d->plan2, d->d_afrag.as<float>(), in_dev, 0, (int64_t)n_frames, (int64_t)n_frames,
                          out_dev, n_out, 0, n_out, (int64_t)n_clips, as_stream(stream));
}

int af_resample_has_stereo_path(uint32_t input_rate, uint32_t output_rate)
{
    if (!input_rate || !output_rate) return 0;
    if (input_rate == output_rate) return 1;
    const uint32_t g = std::gcd(input_rate, output_rate);
    return banded::make_plan(output_rate / g, input_rate / g, rs::kT, rs::kT / 2, 2).ok ? 1 : 0;
}

af_status af_resample_batch(uint32_t input_rate, uint32_t output_rate, const float* in, size_t n_clips,
                            size_t n_in, float* out, size_t out_cap)
{
    AF_REQUIRE(input_rate > 0 && output_rate > 0, "sample rates must be positive");
    const size_t n_out = af_resample_out_len(input_rate, output_rate, n_in);
    if (out_cap < n_out * n_clips)
        return fail(AF_ERR_BUFFER_TOO_SMALL, "resample_batch: need %zu floats", n_out * n_clips);
    if (n_clips == 0 || n_in == 0) return AF_OK;
    AF_REQUIRE(in && out, "af_resample_batch: null buffer");
    AF_TRY(ensure_device(-1));
    // Device staging of this convenience entry point: kept per device between calls -- allocating and freeing a
    // few hundred MB per call costs more than the kernel -- but no more than kKeepBytes of it: what a bigger call
    // needed goes back when the call returns (a config-3-sized call would otherwise pin 10 GB of HBM for the life
    // of the process).  One caller at a time PER DEVICE; af_resample_release_staging() returns the rest.
    const int dev_ix = current_device() < kMaxDevices ? current_device() : 0;
    rs::Staging& stg = rs::staging_slot(dev_ix);
    std::lock_guard<std::mutex> lock(stg.mu);
    DevBuf& din = stg.in;
    DevBuf& dout = stg.out;
    struct Trim {
        rs::Staging& s;
        ~Trim()
        {
            if (s.in.bytes + s.out.bytes > rs::kKeepBytes) { s.in.release(); s.out.release(); }
        }
    } trim{stg};
    AF_TRY(din.reserve(n_clips * n_in * sizeof(float)));
    AF_TRY(dout.reserve(n_clips * n_out * sizeof(float)));
    // Page-locked input of some size: chunks of whole clips (about 32 MB each) with the H2D copy of chunk
    // k + 1, the kernel on chunk k and the D2H copy of chunk k - 1 on three streams -- the output is a third
    // of the input's size or more, so hiding its way back is worth 20-30 % end to end.  Pageable memory goes
    // through the runtime's own staging, which is fastest as ONE copy each way.
    hipPointerAttribute_t pa;
    const bool pinned = hipPointerGetAttributes(&pa, in) == hipSuccess && pa.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();
    const size_t clip_bytes = n_in * sizeof(float);
    size_t per = clip_bytes ? (((size_t)32 << 20) + clip_bytes - 1) / clip_bytes : n_clips;
    if (per < 1) per = 1;
    if (pinned && n_clips * clip_bytes >= ((size_t)64 << 20) && per < n_clips) {
        AF_TRY(af_resample_prepare(input_rate, output_rate));       // tables before the first async launch
        hipStream_t s_in = nullptr, s_run = nullptr, s_out = nullptr;
        std::vector<hipEvent_t> ev;
        auto cleanup = [&]() {
            for (hipEvent_t e : ev) (void)hipEventDestroy(e);
            if (s_in) (void)hipStreamDestroy(s_in);
            if (s_run) (void)hipStreamDestroy(s_run);
            if (s_out) (void)hipStreamDestroy(s_out);
        };
        af_status st = AF_OK;
        auto hip_ok = [&](hipError_t e, const char* what) {
            if (e != hipSuccess && st == AF_OK) st = fail(AF_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
            return e == hipSuccess;
        };
        bool ok = hip_ok(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking), "hipStreamCreate") &&
                  hip_ok(hipStreamCreateWithFlags(&s_run, hipStreamNonBlocking), "hipStreamCreate") &&
                  hip_ok(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking), "hipStreamCreate");
        for (size_t c0 = 0; ok && c0 < n_clips; c0 += per) {
            const size_t nc = n_clips - c0 < per ? n_clips - c0 : per;
            hipEvent_t e_in = nullptr, e_run = nullptr;
            ok = hip_ok(hipEventCreateWithFlags(&e_in, hipEventDisableTiming), "hipEventCreate");
            if (ok) ev.push_back(e_in);
            ok = ok && hip_ok(hipEventCreateWithFlags(&e_run, hipEventDisableTiming), "hipEventCreate");
            if (ok) ev.push_back(e_run);
            ok = ok && hip_ok(hipMemcpyAsync(din.as<float>() + c0 * n_in, in + c0 * n_in, nc * clip_bytes,
                                             hipMemcpyHostToDevice, s_in), "hipMemcpyAsync") &&
                 hip_ok(hipEventRecord(e_in, s_in), "hipEventRecord") &&
                 hip_ok(hipStreamWaitEvent(s_run, e_in, 0), "hipStreamWaitEvent");
            if (ok) {
                const af_status k = af_resample_batch_device(input_rate, output_rate, din.as<float>() + c0 * n_in, nc,
                                                             n_in, dout.as<float>() + c0 * n_out, s_run);
                if (k != AF_OK) { st = k; ok = false; }
            }
            ok = ok && hip_ok(hipEventRecord(e_run, s_run), "hipEventRecord") &&
                 hip_ok(hipStreamWaitEvent(s_out, e_run, 0), "hipStreamWaitEvent") &&
                 hip_ok(hipMemcpyAsync(out + c0 * n_out, dout.as<float>() + c0 * n_out, nc * n_out * sizeof(float),
                                       hipMemcpyDeviceToHost, s_out), "hipMemcpyAsync");
        }
        // drain everything that was enqueued before the buffers and streams go away, error or not
        if (s_in) (void)hipStreamSynchronize(s_in);
        if (s_run) (void)hipStreamSynchronize(s_run);
        if (s_out) hip_ok(hipStreamSynchronize(s_out), "hipStreamSynchronize");
        cleanup();
        return st;
    }
    AF_HIP(hipMemcpy(din.p, in, n_clips * n_in * sizeof(float), hipMemcpyHostToDevice));
    AF_TRY(af_resample_batch_device(input_rate, output_rate, din.as<float>(), n_clips, n_in,
                                    dout.as<float>(), nullptr));
    AF_HIP(hipMemcpy(out, dout.p, n_clips * n_out * sizeof(float), hipMemcpyDeviceToHost));
    return AF_OK;
}

// ---- optional cubic mode (what the reference CONFIGURES: rubato FastFixedIn, PolynomialDegree::Cubic) ----------
// docs/OPS.md "Cubic interpolation": y[m] = the cubic through x[i - 1 .. i + 2] at i + t, i + t = m * in / out in
// float64, zero extension at both ends, out_len = ceil(n L / M).  One thread per output, one clip row per blockIdx.y
// range: 4 B out + 4 B * M / L in per output, HBM-bound.
af_status af_resample_cubic_batch_device(uint32_t input_rate, uint32_t output_rate, const float* in_dev, size_t n_clips,
                                         size_t n_in, float* out_dev, void* stream)
{
    AF_REQUIRE(input_rate > 0 && output_rate > 0, "sample rates must be positive");
    if (n_clips == 0 || n_in == 0) return AF_OK;
    AF_REQUIRE(in_dev && out_dev, "af_resample_cubic_batch_device: null buffer");
    AF_TRY(ensure_device(-1));
    const size_t n_out = af_resample_out_len(input_rate, output_rate, n_in);
    const size_t total = n_clips * n_out;
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, (size_t)1 << 30);
    hipLaunchKernelGGL(rs::cubic_kernel, dim3(grid), dim3(256), 0, as_stream(stream), in_dev, out_dev, (int64_t)n_in,
                       (int64_t)n_out, (int64_t)total, (double)input_rate / (double)output_rate);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

af_status af_resample_cubic_batch(uint32_t input_rate, uint32_t output_rate, const float* in, size_t n_clips, size_t n_in,
                                  float* out, size_t out_cap)
{
    AF_REQUIRE(input_rate > 0 && output_rate > 0, "sample rates must be positive");
    const size_t n_out = af_resample_out_len(input_rate, output_rate, n_in);
    if (out_cap < n_out * n_clips)
        return fail(AF_ERR_BUFFER_TOO_SMALL, "resample_cubic_batch: need %zu floats", n_out * n_clips);
    if (n_clips == 0 || n_in == 0) return AF_OK;
    AF_REQUIRE(in && out, "af_resample_cubic_batch: null buffer");
    AF_TRY(ensure_device(-1));
    DevBuf din, dout;
    AF_TRY(din.reserve(n_clips * n_in * sizeof(float)));
    AF_TRY(dout.reserve(n_clips * n_out * sizeof(float)));
    AF_HIP(hipMemcpy(din.p, in, n_clips * n_in * sizeof(float), hipMemcpyHostToDevice));
    AF_TRY(af_resample_cubic_batch_device(input_rate, output_rate, din.as<float>(), n_clips, n_in, dout.as<float>(), nullptr));
    AF_HIP(hipMemcpy(out, dout.p, n_clips * n_out * sizeof(float), hipMemcpyDeviceToHost));
    return AF_OK;
}

void af_resample_release_staging(void)
{
    for (int d = 0; d < kMaxDevices; ++d) {
        rs::Staging& s = rs::staging_slot(d);
        std::lock_guard<std::mutex> lock(s.mu);
        if (!s.in.p && !s.out.p) continue;
        int cur = 0;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        if (have && cur != d) (void)hipSetDevice(d);
        s.in.release();
        s.out.release();
        if (have && cur != d) (void)hipSetDevice(cur);
    }
}

af_status af_resample_taps(uint32_t input_rate, uint32_t output_rate, float* taps, size_t cap,
                           uint32_t* L, uint32_t* M, uint32_t* T)
{
    rs::Design d;
    AF_TRY(rs::make_design(input_rate, output_rate, d, false));
    if (L) *L = d.L;
    if (M) *M = d.M;
    if (T) *T = rs::kT;
    if (taps) {
        if (cap < d.proto.size()) return fail(AF_ERR_BUFFER_TOO_SMALL, "need %zu floats", d.proto.size());
        std::memcpy(taps, d.proto.data(), d.proto.size() * sizeof(float));
    }
    return AF_OK;
}

}  // extern "C"

// BatchResampler (resampler.rs:114-167): 128-frame chunker over AudioResampler.
struct af_batch_resampler {
    af_resampler* inner = nullptr;
    std::vector<float> buffer;           // resampler.rs:116
    size_t chunk = 128;                  // resampler.rs:55
};

extern "C" {

af_status af_batch_resampler_create(uint32_t input_rate, uint32_t output_rate,
                                    af_batch_resampler** out)
{
    AF_REQUIRE(out, "af_batch_resampler_create: null output");
    *out = nullptr;
    af_batch_resampler* b = new (std::nothrow) af_batch_resampler();
    if (!b) return fail(AF_ERR_OUT_OF_MEMORY, "out of host memory");
    af_status st = af_resampler_create(input_rate, output_rate, &b->inner);
    if (st != AF_OK) { delete b; return st; }
    *out = b;
    return AF_OK;
}

void af_batch_resampler_destroy(af_batch_resampler* b)
{
    if (!b) return;
    af_resampler_destroy(b->inner);
    delete b;
}

size_t af_batch_resampler_buffered(const af_batch_resampler* b) { return b ? b->buffer.size() : 0; }

size_t af_batch_resampler_out_len(const af_batch_resampler* b, size_t n_in)
{
    if (!b) return 0;
    if (b->inner->passthrough) return n_in;
    // process(): whole chunks of carry+n_in; flush(): one padded chunk
    const size_t total = b->buffer.size() + n_in;
    const size_t fed = ((total + b->chunk - 1) / b->chunk) * b->chunk;
    return af_resampler_out_len(b->inner, fed);
}

af_status af_batch_resampler_process(af_batch_resampler* b, const float* in, size_t n_in, float* out,
                                     size_t out_cap, size_t* n_out)
{
    AF_REQUIRE(b, "af_batch_resampler_process: null handle");
    AF_REQUIRE(in || n_in == 0, "af_batch_resampler_process: null input");
    if (b->inner->passthrough)
        // Reference: chunk_size 0 makes `while len >= 0` spin forever
        // (resampler.rs:34-39,:138).  Documented deviation: passthrough.
        return af_resampler_process(b->inner, in, n_in, out, out_cap, n_out);
    b->buffer.insert(b->buffer.end(), in, in + n_in);            // :133
    const size_t n_full = (b->buffer.size() / b->chunk) * b->chunk;   // :138-144, all chunks at once:
    // the inner resampler is streaming, so feeding k chunks one by one or in
    // one call produces the same samples.
    size_t got = 0;
    af_status st = AF_OK;
    if (n_full) {
        st = af_resampler_process(b->inner, b->buffer.data(), n_full, out, out_cap, &got);
        if (st == AF_OK) b->buffer.erase(b->buffer.begin(), b->buffer.begin() + n_full);
        else b->buffer.resize(b->buffer.size() - n_in);          // leave the handle as it was
    }
    if (n_out) *n_out = got;
    return st;
}

af_status af_batch_resampler_flush(af_batch_resampler* b, float* out, size_t out_cap, size_t* n_out)
{
    AF_REQUIRE(b, "af_batch_resampler_flush: null handle");
    if (n_out) *n_out = 0;
    if (b->buffer.empty()) return AF_OK;                          // :153
    if (b->inner->passthrough) { b->buffer.clear(); return AF_OK; }
    std::vector<float> chunk(b->buffer);                          // :156-157 zero-pad to one chunk
    chunk.resize(b->chunk, 0.0f);
    size_t got = 0;
    AF_TRY(af_resampler_process(b->inner, chunk.data(), chunk.size(), out, out_cap, &got));   // :159
    b->buffer.clear();                                            // :162
    if (n_out) *n_out = got;
    return AF_OK;
}

}  // extern "C"
