// fir.hip -- FirFilter (docs/OPS.md "FIR") [B]: causal direct-form FIR,
//   y[n] = sum_{k<K} h[k] x[n-k],  zero initial state, same length as input.
// The reference only carries config flags for such filtering
// (src-tauri/src/modules/config/manager.rs:27-32: noise_suppression/auto_gain)
// and no implementation; the surface follows AudioResampler's style
// (resampler.rs:32,:71): create / process(&mut self) / reset.
//
// Device path: up to 177 taps run as a banded (Toeplitz) matrix product on MFMA
// (banded.hpp).  Longer filters run the VALU kernel below: taps and the input tile
// (+ K-1 halo) live in LDS; a lane produces 5 consecutive outputs from a 5-register
// sliding window, so each tap costs one broadcast LDS read (h[k]), one LDS read (the
// next older sample; lane stride 5 dwords is bank-conflict free) and 5 FMAs.
#include <cstring>
#include <new>
#include <vector>

#include "af_common.hpp"
#include "banded.hpp"

namespace af {
namespace firk {

constexpr int kBlock = 256;
constexpr int kR = 5;                          // outputs per lane
constexpr int kTileOut = kBlock * kR;          // 1280 outputs per tile
constexpr int kMaxTaps = 1024;

struct Geometry {
    const float* x;       // clip 0; x[0] has absolute stream index x_abs0
    float* y;             // clip 0; y[0] is output m0
    const float* taps;    // [K]
    int64_t x_abs0, x_len, x_stride, y_stride;
    int64_t m0, n_out;    // outputs per clip
    int64_t tiles_per_clip;
    int64_t n_tiles;
    int K;
    int Kpad;             // K rounded up to a multiple of 4
};

__global__ __launch_bounds__(kBlock) void fir_kernel(const Geometry g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hs = smem;                     // Kpad taps (zero padded)
    float* xs = smem + g.Kpad;            // Kpad - 1 halo ... + tile
    const int tid = threadIdx.x;
    for (int i = tid; i < g.Kpad; i += kBlock) hs[i] = i < g.K ? g.taps[i] : 0.0f;
    const int halo = g.Kpad - 1;

    for (int64_t tile = blockIdx.x; tile < g.n_tiles; tile += gridDim.x) {
        const int64_t clip = tile / g.tiles_per_clip;
        const int64_t n0 = (tile - clip * g.tiles_per_clip) * kTileOut;   // relative to m0
        const float* xc = g.x + clip * g.x_stride;
        __syncthreads();
        // xs[i] = x[m0 + n0 - halo + i], i in [0, halo + kTileOut); zero outside the buffer
        for (int i = tid; i < halo + kTileOut; i += kBlock) {
            const int64_t rel = g.m0 + n0 - halo + i - g.x_abs0;
            xs[i] = (rel >= 0 && rel < g.x_len) ? xc[rel] : 0.0f;
        }
        __syncthreads();
        const int o0 = tid * kR;                       // first output of this lane in the tile
        const float* xp = xs + halo + o0;              // x[n], n = m0 + n0 + o0
        float win[kR], acc[kR];
#pragma unroll
        for (int j = 0; j < kR; ++j) { win[j] = xp[j]; acc[j] = 0.0f; }
        // win[j] holds x[n + j - k]; after each tap shift in x[n - k - 1]
        for (int k = 0; k < g.Kpad; k += 4) {
            const float4 h4 = *reinterpret_cast<const float4*>(hs + k);
            const float hk[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int j = 0; j < kR; ++j) acc[j] += hk[u] * win[j];
#pragma unroll
                for (int j = kR - 1; j > 0; --j) win[j] = win[j - 1];
                win[0] = xp[-(k + u) - 1];            // >= xs[0] because k+u <= Kpad-1 = halo
            }
        }
        float* yc = g.y + clip * g.y_stride + n0 + o0;
#pragma unroll
        for (int j = 0; j < kR; ++j)
            if (n0 + o0 + j < g.n_out) yc[j] = acc[j];
    }
}

// `plan`/`afrag_dev`: the MFMA path when plan.ok() (afrag built from the same taps).
af_status launch(const banded::FirParts& plan, const float* afrag_dev, const float* taps_dev, int K,
                 const float* x_dev, int64_t x_abs0, int64_t x_len, int64_t x_stride, float* y_dev,
                 int64_t y_stride, int64_t m0, int64_t n_out, int64_t n_clips, hipStream_t stream)
{
    if (n_out <= 0 || n_clips <= 0) return AF_OK;
    if (banded::fits(plan, x_len))
        return banded::launch_fir(plan, afrag_dev, x_dev, x_abs0, x_len, x_stride, y_dev, y_stride, m0,
                                  n_out, n_clips, stream);
    Geometry g{};
    g.x = x_dev;
    g.y = y_dev;
    g.taps = taps_dev;
    g.x_abs0 = x_abs0;
    g.x_len = x_len;
    g.x_stride = x_stride;
    g.y_stride = y_stride;
    g.m0 = m0;
    g.n_out = n_out;
    g.tiles_per_clip = (n_out + kTileOut - 1) / kTileOut;
    g.n_tiles = g.tiles_per_clip * n_clips;
    g.K = K;
    g.Kpad = (K + 3) & ~3;
    const size_t smem = sizeof(float) * ((size_t)g.Kpad + (size_t)(g.Kpad - 1) + kTileOut + 4);
    int64_t grid = g.n_tiles;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(fir_kernel, dim3((unsigned)grid), dim3(kBlock), smem, stream, g);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

}  // namespace firk
}  // namespace af

using namespace af;

struct af_fir {
    std::vector<float> taps;
    std::vector<float> hist;     // K-1 most recent input samples (zeros initially)
    int64_t n_total = 0;         // samples consumed so far (anchors the MFMA tile grid)
    banded::FirParts plan;
    DevBuf d_taps, d_afrag, d_x, d_y;
};

namespace {

// A fragments for af_fir_batch_device(), whose taps arrive as a device pointer: rebuilt on the
// caller's stream before every launch, one scratch buffer per (thread, stream).
struct Scratch {
    hipStream_t stream;
    DevBuf buf;
};
af_status scratch_afrag(hipStream_t stream, size_t floats, float** out)
{
    static thread_local std::vector<Scratch*> pool;
    for (Scratch* s : pool)
        if (s->stream == stream) {
            AF_TRY(s->buf.reserve(floats * sizeof(float)));
            *out = s->buf.as<float>();
            return AF_OK;
        }
    Scratch* s = new (std::nothrow) Scratch();
    if (!s) return fail(AF_ERR_OUT_OF_MEMORY, "out of host memory");
    s->stream = stream;
    af_status st = s->buf.reserve((floats > 48 * 64 ? floats : 48 * 64) * sizeof(float));
    if (st != AF_OK) { delete s; return st; }
    pool.push_back(s);
    *out = s->buf.as<float>();
    return AF_OK;
}

}  // namespace

extern "C" {

af_status af_fir_create(const float* taps, size_t n_taps, af_fir** out)
{
    AF_REQUIRE(out, "af_fir_create: null output");
    *out = nullptr;
    AF_REQUIRE(taps && n_taps >= 1 && n_taps <= (size_t)firk::kMaxTaps,
               "FIR needs 1..%d taps (got %zu)", firk::kMaxTaps, n_taps);
    AF_TRY(ensure_device(-1));
    af_fir* f = new (std::nothrow) af_fir();
    if (!f) return fail(AF_ERR_OUT_OF_MEMORY, "out of host memory");
    f->taps.assign(taps, taps + n_taps);
    f->hist.assign(n_taps - 1, 0.0f);
    f->plan = banded::make_fir_parts((int)n_taps);
    af_status st = f->d_taps.reserve(n_taps * sizeof(float));
    if (st == AF_OK && hipMemcpy(f->d_taps.p, taps, n_taps * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        st = fail(AF_ERR_HIP, "uploading FIR taps failed");
    if (st == AF_OK && f->plan.ok()) {
        st = f->d_afrag.reserve(f->plan.afrag_floats * sizeof(float));
        if (st == AF_OK) st = banded::build_fir_afrag(f->plan, f->d_taps.as<float>(), f->d_afrag.as<float>(), nullptr);
        if (st == AF_OK && hipStreamSynchronize(nullptr) != hipSuccess)
            st = fail(AF_ERR_HIP, "building the FIR fragments failed");
    }
    if (st != AF_OK) { delete f; return st; }
    *out = f;
    return AF_OK;
}

void af_fir_destroy(af_fir* f) { delete f; }

void af_fir_reset(af_fir* f)
{
    if (!f) return;
    std::fill(f->hist.begin(), f->hist.end(), 0.0f);
    f->n_total = 0;
}

af_status af_fir_process(af_fir* f, const float* in, size_t n, float* out)
{
    AF_REQUIRE(f, "af_fir_process: null handle");
    if (n == 0) return AF_OK;
    AF_REQUIRE(in && out, "af_fir_process: null buffer");
    const size_t K = f->taps.size(), h = K - 1;
    AF_TRY(f->d_x.reserve((h + n) * sizeof(float)));
    AF_TRY(f->d_y.reserve(n * sizeof(float)));
    if (h) AF_HIP(hipMemcpy(f->d_x.p, f->hist.data(), h * sizeof(float), hipMemcpyHostToDevice));
    AF_HIP(hipMemcpy(f->d_x.as<float>() + h, in, n * sizeof(float), hipMemcpyHostToDevice));
    AF_TRY(firk::launch(f->plan, f->d_afrag.as<float>(), f->d_taps.as<float>(), (int)K, f->d_x.as<float>(),
                        f->n_total - (int64_t)h, (int64_t)(h + n), 0, f->d_y.as<float>(), 0, f->n_total,
                        (int64_t)n, 1, nullptr));
    AF_HIP(hipMemcpy(out, f->d_y.p, n * sizeof(float), hipMemcpyDeviceToHost));
    // carry the K-1 most recent samples
    if (h) {
        if (n >= h) {
            f->hist.assign(in + (n - h), in + n);
        } else {
            f->hist.erase(f->hist.begin(), f->hist.begin() + n);
            f->hist.insert(f->hist.end(), in, in + n);
        }
    }
    f->n_total += (int64_t)n;
    return AF_OK;
}

af_status af_fir_apply_batch_device(const af_fir* f, const float* in_dev, size_t n_clips, size_t n,
                                    float* out_dev, void* stream)
{
    AF_REQUIRE(f, "af_fir_apply_batch_device: null handle");
    if (n_clips == 0 || n == 0) return AF_OK;
    AF_REQUIRE(in_dev && out_dev, "af_fir_apply_batch_device: null buffer");
    return firk::launch(f->plan, f->d_afrag.as<float>(), f->d_taps.as<float>(), (int)f->taps.size(), in_dev, 0,
                        (int64_t)n, (int64_t)n, out_dev, (int64_t)n, 0, (int64_t)n, (int64_t)n_clips,
                        as_stream(stream));
}

af_status af_fir_batch_device(const float* taps_dev, size_t n_taps, const float* in_dev, size_t n_clips,
                              size_t n, float* out_dev, void* stream)
{
    AF_REQUIRE(n_taps >= 1 && n_taps <= (size_t)firk::kMaxTaps, "FIR needs 1..%d taps (got %zu)",
               firk::kMaxTaps, n_taps);
    if (n_clips == 0 || n == 0) return AF_OK;
    AF_REQUIRE(taps_dev && in_dev && out_dev, "af_fir_batch_device: null buffer");
    const banded::FirParts plan = banded::make_fir_parts((int)n_taps);
    float* afrag = nullptr;
    if (plan.ok()) {
        AF_TRY(scratch_afrag(as_stream(stream), plan.afrag_floats, &afrag));
        AF_TRY(banded::build_fir_afrag(plan, taps_dev, afrag, as_stream(stream)));
    }
    return firk::launch(plan, afrag, taps_dev, (int)n_taps, in_dev, 0, (int64_t)n, (int64_t)n, out_dev,
                        (int64_t)n, 0, (int64_t)n, (int64_t)n_clips, as_stream(stream));
}

af_status af_fir_batch(const float* taps, size_t n_taps, const float* in, size_t n_clips, size_t n,
                       float* out)
{
    AF_REQUIRE(n_taps >= 1 && n_taps <= (size_t)firk::kMaxTaps, "FIR needs 1..%d taps (got %zu)",
               firk::kMaxTaps, n_taps);
    if (n_clips == 0 || n == 0) return AF_OK;
    AF_REQUIRE(taps && in && out, "af_fir_batch: null buffer");
    AF_TRY(ensure_device(-1));
    DevBuf dt, din, dout;
    AF_TRY(dt.reserve(n_taps * sizeof(float)));
    AF_TRY(din.reserve(n_clips * n * sizeof(float)));
    AF_TRY(dout.reserve(n_clips * n * sizeof(float)));
    AF_HIP(hipMemcpy(dt.p, taps, n_taps * sizeof(float), hipMemcpyHostToDevice));
    AF_HIP(hipMemcpy(din.p, in, n_clips * n * sizeof(float), hipMemcpyHostToDevice));
    AF_TRY(af_fir_batch_device(dt.as<float>(), n_taps, din.as<float>(), n_clips, n, dout.as<float>(),
                               nullptr));
    AF_HIP(hipMemcpy(out, dout.p, n_clips * n * sizeof(float), hipMemcpyDeviceToHost));
    return AF_OK;
}

}  // extern "C"
