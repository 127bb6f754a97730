// banded_mfma.hip -- see banded.hpp.  One workgroup = nwaves waves; wave w owns block matrix
// p = w / wpp (its NK A fragments live in registers for the whole launch) and the block
// periods g in [gsub * gpw, (gsub + 1) * gpw) of every column.  A tile is 16 columns, each a
// private LDS segment of G block periods of x (+ the window halo) with an odd stride, so the
// 16 columns x 2 k-groups of a ds_read_b32 lane group fall into 32 different banks.
//
// k-slot mapping: MFMA step t, k-group g4 = lane / 16 reads window sample
//   w(t, g4) = 64 (t / 16) + t % 16 + 16 g4      for the NK / 16 full 16-step chunks,
//   w(t, g4) = 64 FULL + (t - 16 FULL) + TAILQ g4  for the TAILQ = NK % 16 tail steps,
// i.e. the two k-groups of a lane group are 16 samples apart (bank-conflict free with an odd
// column stride).  The A fragments are gathered with the same mapping (build_afrag_kernel).
#include <numeric>
#include <type_traits>

#include "banded.hpp"

namespace af {
namespace banded {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kCols = 16;
constexpr int kPfSlots = 2;          // x prefetch registers per column and thread
constexpr int kMaxSeg = 1000;
constexpr int64_t kMaxClip = (1ll << 30) - (1ll << 20);   // byte offsets of a clip + one tile < 2^32        // floats per column segment (16 segments <= 64 KB)

struct Geom {
    const float* x;
    float* y;
    const float* afrag;
    int64_t x_abs0, x_len, x_stride, y_stride, m0, n_out;
    int64_t tile0;                   // absolute index of the first tile of every clip
    int64_t tiles_per_clip, n_tiles;
    uint32_t L, M;
    int lo_off;                      // c - (T - 1): first window sample relative to q(16 b)
    int nb, S, G, wpp, gpw, seg_len, seg_stride;
    int accumulate;                  // 1: y += (a later pass of a long FIR)
};

__device__ __forceinline__ int w_of(int t, int g4, int nk)
{
    const int full = nk / 16, tailq = nk % 16;
    return t < 16 * full ? 64 * (t / 16) + (t % 16) + 16 * g4 : 64 * full + (t - 16 * full) + tailq * g4;
}

__global__ void build_afrag_kernel(const float* __restrict__ proto, float* __restrict__ afrag,
                                   uint32_t L, uint32_t M, int T, int nb, int nk)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nb * nk * 64) return;
    const int lane = idx & 63, t = (idx >> 6) % nk, p = (idx >> 6) / nk;
    const int i = lane & 15, g4 = lane >> 4;
    const uint64_t r0 = ((uint64_t)16 * p * M) % L;
    const uint64_t pos = r0 + (uint64_t)i * M;           // = r_i + d_i L
    const int d = (int)(pos / L);
    const uint32_t r = (uint32_t)(pos % L);
    const int k = d + (T - 1) - w_of(t, g4, nk);
    afrag[idx] = (k >= 0 && k < T) ? proto[(size_t)r + (size_t)k * L] : 0.0f;
}

// A zero the compiler cannot see through: offsets derived from it are recomputed where they
// are used instead of being hoisted out of the tile loop into 30+ live registers.
__device__ __forceinline__ int opaque_zero()
{
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

// WPE = waves per SIMD the register budget is cut for: 5 when two workgroups of 9-10 waves
// share a CU (96 VGPRs), 4 otherwise (128).
// CH = 2: x is interleaved stereo and is downmixed on its way into LDS ((L + R) * 0.5, the
// to_mono of capture.rs:30-42): the staging registers hold (L, R) pairs.
template <int NK, int WPE, int CH>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void banded_kernel(const Geom g)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];
    constexpr int FULL = NK / 16, TAILQ = NK % 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = wave / g.wpp, gsub = wave - p * g.wpp;
    const int qp = (int)(((int64_t)16 * p * g.M) / g.L);      // q(16 p)

    float a[NK];
#pragma unroll
    for (int t = 0; t < NK; ++t) a[t] = g.afrag[((size_t)p * NK + t) * 64 + lane];

    const int col = lane & 15, g4 = lane >> 4;
    const int wave_off = qp + gsub * g.gpw * g.S;
    const int base_full = col * g.seg_stride + 16 * g4 + wave_off;
    const int base_tail = col * g.seg_stride + TAILQ * g4 + 64 * FULL + wave_off;
    // this lane's outputs: rows 4 g4 .. 4 g4 + 3 of block (p, column col, period g)
    const int out_lane = 16 * (p + g.nb * (col * g.G + gsub * g.gpw)) + 4 * g4;
    const int out_step = 16 * g.nb;                             // per block period
    const int out_per_tile = 256 * g.G * g.nb;
    const int col_span = g.G * g.S;
    // seg_len may be shorter than the workgroup (short block periods, e.g. 48 k -> 45 k): lanes
    // past the segment neither load nor store, or they would land in the neighbouring column
    const bool first = tid < g.seg_len;
    const bool second = tid + nthreads < g.seg_len;             // this thread fills two samples per segment
    const uint32_t tpc = (uint32_t)g.tiles_per_clip, n_tiles = (uint32_t)g.n_tiles;

    // x comes through a buffer resource over the clip: samples before / after the clip read as
    // zero in hardware, the byte offset wraps modulo 2^32 (launch() bounds x_len for that).
    // The 32 loads of the next tile are issued one by one between the MFMAs of the first chain
    // pair, so no wave sits in a burst of address work while the matrix pipe idles.
    constexpr uint32_t kOob = 0xFFF00000u;                      // + 16 col_span * 4 stays out of range
    float pf[kCols * kPfSlots * CH];
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[kPfSlots];
    auto prep_tile = [&](uint32_t tile, bool live) {
        const uint32_t clip = tile / tpc;
        const int64_t t_abs = g.tile0 + (tile - clip * tpc);
        const int64_t base = t_abs * (16 * (int64_t)col_span) + g.lo_off - g.x_abs0;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x + (int64_t)clip * g.x_stride * CH), 0,
                                                 (int)(uint32_t)(g.x_len * 4 * CH), 0x00020000);
        const uint32_t v0 = ((uint32_t)base + (uint32_t)(tid + opaque_zero())) * (4u * CH);
        voff[0] = live && first ? v0 : kOob;
        voff[1] = live && second ? v0 + (uint32_t)nthreads * (4u * CH) : kOob;
    };
    auto issue_load = [&](int s) {                              // s = h * 16 + cc
        const int cc = s % kCols, h = s / kCols;
        if constexpr (CH == 1) {
            pf[cc * kPfSlots + h] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(voff[h] + (uint32_t)(cc * col_span) * 4u), 0, 0));
        } else {
            // two dword loads: hipcc 7.2 lowers __builtin_amdgcn_raw_buffer_load_b64 to a single
            // dword load and duplicates it
            const int off = (int)(voff[h] + (uint32_t)(cc * col_span) * 8u);
            pf[(cc * kPfSlots + h) * 2] =
                __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
            pf[(cc * kPfSlots + h) * 2 + 1] =
                __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off + 4, 0, 0));
        }
    };
    constexpr int kLoads = kCols * kPfSlots;

    uint32_t tile = blockIdx.x;
    if (tile < n_tiles) {
        prep_tile(tile, true);
#pragma unroll
        for (int s2 = 0; s2 < kLoads; ++s2) issue_load(s2);
    }
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                                        // readers of the previous tile done
        {
            float* dst = xs + tid + opaque_zero();
            auto staged = [&](int cc, int h) {                  // downmix: (L + R) * 0.5, exact
                if constexpr (CH == 1) return pf[cc * kPfSlots + h];
                else return (pf[(cc * kPfSlots + h) * 2] + pf[(cc * kPfSlots + h) * 2 + 1]) * 0.5f;
            };
            if (first) {
#pragma unroll
                for (int cc = 0; cc < kCols; ++cc) dst[cc * g.seg_stride] = staged(cc, 0);
            }
            if (second) {
#pragma unroll
                for (int cc = 0; cc < kCols; ++cc) dst[cc * g.seg_stride + nthreads] = staged(cc, 1);
            }
        }
        __syncthreads();
        const uint32_t clip = tile / tpc;
        const int64_t t_abs = g.tile0 + (tile - clip * tpc);
        {
            const bool live = tile + gridDim.x < n_tiles;
            prep_tile(live ? tile + gridDim.x : tile, live);
        }

        const int64_t mbase = t_abs * out_per_tile - g.m0;      // output index of the tile - m0
        float* yc = g.y + (int64_t)clip * g.y_stride + mbase;
        const bool whole = mbase >= 0 && mbase + out_per_tile <= g.n_out &&
                           ((reinterpret_cast<uintptr_t>(yc) >> 2) & 3) == 0;
        const int lo = mbase >= 0 ? 0 : -mbase > out_per_tile ? out_per_tile : (int)-mbase;
        const int64_t left = g.n_out - mbase;
        const int hi = left < 0 ? 0 : left > out_per_tile ? out_per_tile : (int)left;

        auto chain_pair = [&](int gi, auto with_loads) {
            constexpr bool LOADS = decltype(with_loads)::value;
            const int off0 = gi * g.S;
            const float* b0p = xs + base_full + off0;
            const float* b1p = b0p + g.S;
            const float* t0p = xs + base_tail + off0;
            const float* t1p = t0p + g.S;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const int o0 = out_lane + gi * out_step, o1 = o0 + out_step;
            if (g.accumulate) {                                 // wave-uniform: start from the earlier passes' sum
                if (whole) {
                    acc0 = *reinterpret_cast<const f32x4*>(yc + o0);
                    acc1 = *reinterpret_cast<const f32x4*>(yc + o1);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (o0 + j >= lo && o0 + j < hi) acc0[j] = yc[o0 + j];
                        if (o1 + j >= lo && o1 + j < hi) acc1[j] = yc[o1 + j];
                    }
                }
            }
            // hand-pipelined: the window samples of step t + kAhead are read before the MFMAs of
            // step t, one prefetch load goes out per step; sched_barrier keeps hipcc from
            // clustering the loads at the end of the chain
            constexpr int kAhead = 3;
            float w0[kAhead + 1], w1[kAhead + 1];            // rotating operand registers
            auto read_b = [&](int t) {
                w0[t % (kAhead + 1)] = t < 16 * FULL ? b0p[64 * (t / 16) + (t % 16)] : t0p[t - 16 * FULL];
                w1[t % (kAhead + 1)] = t < 16 * FULL ? b1p[64 * (t / 16) + (t % 16)] : t1p[t - 16 * FULL];
            };
#pragma unroll
            for (int t = 0; t < kAhead && t < NK; ++t) read_b(t);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NK; ++t) {
                if (t + kAhead < NK) read_b(t + kAhead);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], w0[t % (kAhead + 1)], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], w1[t % (kAhead + 1)], acc1, 0, 0, 0);
                if (LOADS) {
                    // all of them in the first half of the chain: the later a load goes out, the
                    // less of the chain is left to cover its latency
                    constexpr int kSpan = (NK + 1) / 2;
                    if (t < kSpan) {
#pragma unroll
                        for (int s2 = (kLoads * t) / kSpan; s2 < (kLoads * (t + 1)) / kSpan; ++s2) issue_load(s2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (whole) {
                *reinterpret_cast<f32x4*>(yc + o0) = acc0;
                *reinterpret_cast<f32x4*>(yc + o1) = acc1;
            } else {                                            // clip edges, unaligned streams
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (o0 + j >= lo && o0 + j < hi) yc[o0 + j] = acc0[j];
                    if (o1 + j >= lo && o1 + j < hi) yc[o1 + j] = acc1[j];
                }
            }
        };
        chain_pair(0, std::true_type{});
        for (int gi = 2; gi < g.gpw; gi += 2) chain_pair(gi, std::false_type{});
    }
}

const int kNkInstances[] = {8, 12, 16, 20, 24, 28, 32, 36, 40, 48};

template <int NK>
void launch_nk(const Geom& g, bool five, int ch, unsigned grid, unsigned block, size_t smem, hipStream_t s)
{
    if constexpr (NK <= 32) {
        if (ch == 2) {
            hipLaunchKernelGGL((banded_kernel<NK, 4, 2>), dim3(grid), dim3(block), smem, s, g);
            return;
        }
        if (five) {
            hipLaunchKernelGGL((banded_kernel<NK, 5, 1>), dim3(grid), dim3(block), smem, s, g);
            return;
        }
    }
    hipLaunchKernelGGL((banded_kernel<NK, 4, 1>), dim3(grid), dim3(block), smem, s, g);
}

}  // namespace

Plan make_plan(uint32_t L, uint32_t M, int T, int c, int ch)
{
    Plan p;
    p.ch = ch;
    p.L = L;
    p.M = M;
    p.T = T;
    p.c = c;
    if (L == 0 || M == 0 || T < 1) return p;
    if (debug_knob(kKnobBandedOff)) return p;               // diagnostics: force the VALU kernels
    const uint32_t g16 = std::gcd(16u, L);
    const uint64_t nb = L / g16;
    const uint64_t S = (uint64_t)(16 / g16) * M;
    if (nb > 16 || S > (uint64_t)kMaxSeg) return p;
    p.nb = (int)nb;
    p.S = (int)S;
    uint64_t d15 = 0;
    for (uint64_t b = 0; b < nb; ++b) {
        const uint64_t r0 = (16 * b * M) % L;
        const uint64_t d = (r0 + 15ull * M) / L;
        if (d > d15) d15 = d;
    }
    const uint64_t W = (uint64_t)T + d15;
    const uint64_t need = 16 * (W / 64) + ((W % 64) + 3) / 4;
    for (int nk : kNkInstances)
        if ((uint64_t)nk >= need) { p.NK = nk; break; }
    if (!p.NK) return p;
    p.wpp = nb >= 5 ? 1 : nb == 1 ? 8 : nb == 2 ? 4 : 2;
    p.nwaves = p.nb * p.wpp;
    const int nthreads = 64 * p.nwaves;
    if (ch == 2 && (p.nwaves > 8 || p.NK > 32)) return p;     // stereo staging: 8-wave shapes only
    const int seg_cap = kPfSlots * nthreads < kMaxSeg ? kPfSlots * nthreads : kMaxSeg;
    const int qp_last = (int)((16ull * (nb - 1) * M) / L);
    const int unit = 2 * p.wpp;                              // every wave runs chains in pairs
    int G = 0;
    for (int cand = unit; (int64_t)(cand - 1) * p.S + qp_last + 4 * p.NK <= seg_cap; cand += unit) G = cand;
    if (!G) return p;
    p.G = G;
    p.seg_len = (G - 1) * p.S + qp_last + 4 * p.NK;
    p.seg_stride = p.seg_len | 1;
    p.out_per_tile = (int64_t)256 * G * p.nb;
    p.smem = (size_t)kCols * p.seg_stride * sizeof(float);
    p.ok = true;
    return p;
}

bool fits(const Plan& p, int64_t x_len) { return p.ok && x_len * p.ch < (int64_t)kMaxClip; }

af_status build_afrag(const Plan& p, const float* proto_dev, float* afrag_dev, hipStream_t stream)
{
    const int n = (int)p.afrag_floats();
    hipLaunchKernelGGL(build_afrag_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, proto_dev,
                       afrag_dev, p.L, p.M, p.T, p.nb, p.NK);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

af_status launch(const Plan& p, const float* afrag_dev, const float* x_dev, int64_t x_abs0,
                 int64_t x_len, int64_t x_stride, float* y_dev, int64_t y_stride, int64_t m0,
                 int64_t n_out, int64_t n_clips, hipStream_t stream, bool accumulate)
{
    if (n_out <= 0 || n_clips <= 0) return AF_OK;
    if (!p.ok) return fail(AF_ERR_UNSUPPORTED, "banded::launch without a plan");
    if (x_len * p.ch >= (int64_t)kMaxClip)
        return fail(AF_ERR_UNSUPPORTED, "clips of %lld samples are beyond the banded kernel's 32-bit offsets",
                    (long long)x_len);
    Geom g{};
    g.x = x_dev;
    g.y = y_dev;
    g.afrag = afrag_dev;
    g.x_abs0 = x_abs0;
    g.x_len = x_len;
    g.x_stride = x_stride;
    g.y_stride = y_stride;
    g.m0 = m0;
    g.n_out = n_out;
    g.tile0 = m0 / p.out_per_tile;
    g.tiles_per_clip = (m0 + n_out - 1) / p.out_per_tile - g.tile0 + 1;
    g.n_tiles = g.tiles_per_clip * n_clips;
    if (g.n_tiles >= (1ll << 31)) return fail(AF_ERR_UNSUPPORTED, "banded::launch: too many tiles");
    g.L = p.L;
    g.M = p.M;
    g.lo_off = p.c - (p.T - 1);
    g.nb = p.nb;
    g.S = p.S;
    g.G = p.G;
    g.wpp = p.wpp;
    g.gpw = p.G / p.wpp;
    g.seg_len = p.seg_len;
    g.seg_stride = p.seg_stride;
    g.accumulate = accumulate ? 1 : 0;
    const int wg_per_cu = (2 * 64 * p.nwaves <= 2048 && 2 * p.smem <= 150 * 1024) ? 2 : 1;
    int64_t grid = g.n_tiles;
    const int64_t n_cus = device_cu_count(current_device());
    if (grid > n_cus * wg_per_cu) grid = n_cus * wg_per_cu;
    const unsigned block = 64u * p.nwaves;
    const bool five = wg_per_cu == 2 && 2 * p.nwaves > 16;
    switch (p.NK) {
    case 8: launch_nk<8>(g, five, p.ch, (unsigned)grid, block, p.smem, stream); break;
    case 12: launch_nk<12>(g, five, p.ch, (unsigned)grid, block, p.smem, stream); break;
    case 16: launch_nk<16>(g, five, p.ch, (unsigned)grid, block, p.smem, stream); break;
    case 20: launch_nk<20>(g, five, p.ch, (unsigned)grid, block, p.smem, stream); break;
    case 24: launch_nk<24>(g, five, p.ch, (unsigned)grid, block, p.smem, stream); break;
    case 28: launch_nk<28>(g, five, p.ch, (unsigned)grid, block, p.smem, stream); break;
    case 32: launch_nk<32>(g, five, p.ch, (unsigned)grid, block, p.smem, stream); break;
    case 36: launch_nk<36>(g, five, p.ch, (unsigned)grid, block, p.smem, stream); break;
    case 40: launch_nk<40>(g, five, p.ch, (unsigned)grid, block, p.smem, stream); break;
    case 48: launch_nk<48>(g, five, p.ch, (unsigned)grid, block, p.smem, stream); break;
    default: return fail(AF_ERR_UNSUPPORTED, "no banded kernel for NK=%d", p.NK);
    }
    AF_HIP(hipGetLastError());
    return AF_OK;
}

FirParts make_fir_parts(int K)
{
    FirParts fp;
    if (K < 1) return fp;
    const Plan whole = make_plan(1, 1, K, 0);
    if (whole.ok) {
        fp.n = 1;
        fp.plan[0] = whole;
        fp.afrag_floats = whole.afrag_floats();
        return fp;
    }
    const int n = (K + kFirPartTaps - 1) / kFirPartTaps;
    if (n > kFirMaxParts) return fp;
    for (int j = 0; j < n; ++j) {
        const int t0 = j * kFirPartTaps, tj = K - t0 < kFirPartTaps ? K - t0 : kFirPartTaps;
        fp.plan[j] = make_plan(1, 1, tj, -t0);               // taps [t0, t0 + tj) see x delayed by t0
        if (!fp.plan[j].ok) return FirParts{};
        fp.tap0[j] = t0;
        fp.afrag_off[j] = fp.afrag_floats;
        fp.afrag_floats += fp.plan[j].afrag_floats();
    }
    fp.n = n;
    return fp;
}

bool fits(const FirParts& p, int64_t x_len) { return p.ok() && fits(p.plan[0], x_len); }

af_status build_fir_afrag(const FirParts& p, const float* taps_dev, float* afrag_dev, hipStream_t stream)
{
    for (int j = 0; j < p.n; ++j)
        AF_TRY(build_afrag(p.plan[j], taps_dev + p.tap0[j], afrag_dev + p.afrag_off[j], stream));
    return AF_OK;
}

af_status launch_fir(const FirParts& p, const float* afrag_dev, const float* x_dev, int64_t x_abs0, int64_t x_len,
                     int64_t x_stride, float* y_dev, int64_t y_stride, int64_t m0, int64_t n_out,
                     int64_t n_clips, hipStream_t stream)
{
    for (int j = 0; j < p.n; ++j)
        AF_TRY(launch(p.plan[j], afrag_dev + p.afrag_off[j], x_dev, x_abs0, x_len, x_stride, y_dev, y_stride, m0,
                      n_out, n_clips, stream, j > 0));
    return AF_OK;
}

}  // namespace banded
}  // namespace af
