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
#include "pcm_dev.hpp"

// AF_BANDED_PAIRS (0: ablation): mono x is staged in sample pairs (8-byte buffer loads) instead of single dwords
#ifndef AF_BANDED_PAIRS
#define AF_BANDED_PAIRS 1
#endif
// cache policy bits of the x requests / the 16-byte y stores (experiment: 2 = nt, once-touched data)
#ifndef AF_X_AUX
#define AF_X_AUX 0
#endif
#ifndef AF_Y_AUX
#define AF_Y_AUX 0
#endif
// AF_BANDED_ABL (diagnostic builds only, scripts/ablate/): parts of banded_kernel compiled OUT: 1 x requests, 2 stores,
// 64 the chains' arithmetic (LDS reads + MFMA).  Results are garbage; bounds stay checked.
#ifndef AF_BANDED_ABL
#define AF_BANDED_ABL 0
#endif
// AF_BANDED_BS (0: ablation): the ten-wave shapes store through a range-checked buffer resource
#ifndef AF_BANDED_BS
#define AF_BANDED_BS 1
#endif

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
    int buffer_stores;               // the ten-wave shapes: y through a range-checked buffer resource (BS)
};

__device__ __forceinline__ int w_of(int t, int g4, int nk)
{
    const int full = nk / 16, tailq = nk % 16;
    return t < 16 * full ? 64 * (t / 16) + (t % 16) + 16 * g4 : 64 * full + (t - 16 * full) + tailq * g4;
}

// quads: the streamed kernel's table, [nb][NK / 4][64][4] (a lane's four consecutive steps are one 16-byte load)
__global__ void build_afrag_kernel(const float* __restrict__ proto, float* __restrict__ afrag,
                                   uint32_t L, uint32_t M, int T, int nb, int nk, int quads)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nb * nk * 64) return;
    int lane = idx & 63, t = (idx >> 6) % nk;
    const int p = (idx >> 6) / nk;
    if (quads) {                                   // idx = ((p * nk / 4 + t4) * 64 + lane) * 4 + j
        const int in_p = idx - p * nk * 64;
        t = 4 * (in_p / 256) + (in_p & 3);
        lane = (in_p >> 2) & 63;
    }
    const int i = lane & 15, g4 = lane >> 4;
    const uint64_t r0 = ((uint64_t)16 * p * M) % L;
    const uint64_t pos = r0 + (uint64_t)i * M;           // = r_i + d_i L
    const int d = (int)(pos / L);
    const uint32_t r = (uint32_t)(pos % L);
    const int k = d + (T - 1) - w_of(t, g4, nk);
    afrag[idx] = (k >= 0 && k < T) ? proto[(size_t)r + (size_t)k * L] : 0.0f;
}

__global__ void build_qtab_kernel(int* __restrict__ qtab, uint32_t L, uint32_t M, int nb)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p <= nb) qtab[p] = (int)(((uint64_t)16 * p * M) / L);       // q(16 p); q(16 nb) = S
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
// ACC = 1: the accumulators start from y (a later pass of a long FIR).  A template parameter on purpose: as a
// runtime branch its loads made hipcc's wait-count pass put s_waitcnt vmcnt(1) in front of every tile's first
// MFMA -- a wait for the previous tile's output stores.
// BS = true (the ten-wave shapes, where launch() may): y goes out through a buffer resource over the wanted outputs
// [m0, m0 + n_out) of the clip, one 16-byte store per chain whatever the tile's position -- the range check drops what
// lies outside, dword by dword.  With predicated stores the number of vector memory operations behind the x requests
// depends on the path, and the wait in front of the staging (one counter for loads and stores, retired in order) had to
// be vmcnt(0): every tile began by waiting for the previous tile's last store to be acknowledged.
template <int NK, int WPE, int CH, bool ACC = false, bool BS = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void banded_kernel(const Geom g)
{
    static_assert(!(BS && (ACC || WPE != 5)), "buffer stores: the one-chain-at-a-time shapes, plain passes");
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

    // PAIR (mono input): a thread stages the sample pair (2 tid, 2 tid + 1) of every column segment with ONE 8-byte
    // load -- half the vector-memory instructions of a tile -- on a pair grid anchored at an even sample of the clip
    // (a pair that straddled sample 0 would have a wrapped byte offset and be dropped whole by the range check): a
    // tile whose first sample is odd starts its pairs one sample early (sh), segments keep one lead-in word for it.
    // col_span = G S is even (G is), so every column of a tile has the same parity.
    // (Not in the five-waves-per-SIMD shapes: config 3's ten-wave workgroup measured 2.21 ms with pairs against 2.12
    // without on one box, while the eight-wave shapes gain 1-3 %; banded.hpp Plan::pairs tells the host which it is.)
    constexpr bool PAIR = CH == 1 && WPE != 5 && AF_BANDED_PAIRS;
    constexpr int kLead = PAIR ? 1 : 0;
    const int col = lane & 15, g4 = lane >> 4;
    const int wave_off = qp + gsub * g.gpw * g.S + kLead;
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
    int sh = 0;                                                 // PAIR: of the tile whose x is in flight
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
        if constexpr (PAIR) {
            sh = (int)(base & 1);
            const uint32_t v0 = ((uint32_t)(base - sh) + 2u * (uint32_t)(tid + opaque_zero())) * 4u;
            voff[0] = live && 2 * tid < g.seg_len + sh ? v0 : kOob;
        } else {
            const uint32_t v0 = ((uint32_t)base + (uint32_t)(tid + opaque_zero())) * (4u * CH);
            voff[0] = live && first ? v0 : kOob;
            voff[1] = live && second ? v0 + (uint32_t)nthreads * (4u * CH) : kOob;
        }
    };
    auto issue_load = [&](int s) {                              // s = h * 16 + cc  (PAIR: s = cc)
        if constexpr (AF_BANDED_ABL & 1) return;
        const int cc = s % kCols, h = s / kCols;
        if constexpr (PAIR) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            // (everything in the VECTOR offset: the scalar offset operand is added behind the range check)
            const f32x2 v = __builtin_bit_cast(
                f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(voff[0] + (uint32_t)(s * col_span) * 4u), 0, AF_X_AUX));
            pf[s * kPfSlots] = v[0];
            pf[s * kPfSlots + 1] = v[1];
        } else if constexpr (CH == 1) {
            pf[cc * kPfSlots + h] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(voff[h] + (uint32_t)(cc * col_span) * 4u), 0, AF_X_AUX));
        } else {
            // one 8-byte load per frame (cast the whole vector: hipcc 7.2 reads element 0 for
            // __builtin_bit_cast(float, v[i]))
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const int off = (int)(voff[h] + (uint32_t)(cc * col_span) * 8u);
            const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, AF_X_AUX));
            pf[(cc * kPfSlots + h) * 2] = v[0];
            pf[(cc * kPfSlots + h) * 2 + 1] = v[1];
        }
    };
    constexpr int kLoads = PAIR ? kCols : kCols * kPfSlots;

    uint32_t tile = blockIdx.x;
    if (tile < n_tiles) {
        prep_tile(tile, true);
#pragma unroll
        for (int s2 = 0; s2 < kLoads; ++s2) issue_load(s2);
        if constexpr (BS) {
            // one store to nowhere (an offset the range check drops) behind the first tile's x requests: every later
            // tile's requests are followed by at least one store, and with this one the compiler's count of "operations
            // younger than the requests" is 1 on every path into the loop instead of 0 -- the staging waits with
            // vmcnt(1) and leaves the previous tile's last store in flight
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t nowhere = __builtin_amdgcn_make_buffer_rsrc(g.y, 0, 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, nowhere, (int)kOob, 0, 0);
        }
    }
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                                        // readers of the previous tile done
        {
            float* dst = xs + tid + opaque_zero();
            auto staged = [&](int cc, int h) {                  // downmix: (L + R) * 0.5, exact
                if constexpr (CH == 1) return pf[cc * kPfSlots + h];
                else return (pf[(cc * kPfSlots + h) * 2] + pf[(cc * kPfSlots + h) * 2 + 1]) * 0.5f;
            };
            if constexpr (PAIR) {
                if (2 * tid < g.seg_len + sh) {
                    float* d2 = xs + 1 - sh + 2 * (tid + opaque_zero());
#pragma unroll
                    for (int cc = 0; cc < kCols; ++cc) {
                        d2[cc * g.seg_stride] = pf[cc * kPfSlots];
                        d2[cc * g.seg_stride + 1] = pf[cc * kPfSlots + 1];
                    }
                }
            } else {
                if (first) {
#pragma unroll
                    for (int cc = 0; cc < kCols; ++cc) dst[cc * g.seg_stride] = staged(cc, 0);
                }
                if (second) {
#pragma unroll
                    for (int cc = 0; cc < kCols; ++cc) dst[cc * g.seg_stride + nthreads] = staged(cc, 1);
                }
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
        // BS: 32 bits of mbase are enough -- the tile grid reaches less than one tile past either end of the wanted outputs
        // and launch() bounds n_out, so no offset wraps INTO the range the hardware accepts
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
            g.y + (int64_t)clip * g.y_stride, 0, (int)(uint32_t)(g.n_out * 4), 0x00020000);
        const uint32_t mb32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mbase);
        const bool whole = mbase >= 0 && mbase + out_per_tile <= g.n_out &&
                           ((reinterpret_cast<uintptr_t>(yc) >> 2) & 3) == 0;
        const int lo = mbase >= 0 ? 0 : -mbase > out_per_tile ? out_per_tile : (int)-mbase;
        const int64_t left = g.n_out - mbase;
        const int hi = left < 0 ? 0 : left > out_per_tile ? out_per_tile : (int)left;

        // Two block periods per call (two independent accumulator chains) where the register budget allows it;
        // the five-waves-per-SIMD shapes (96 VGPRs) run ONE chain at a time: with pairs hipcc spilled staging
        // registers right behind their buffer_load, i.e. an s_waitcnt vmcnt(0) on an HBM load in every tile.
        constexpr bool kPair = WPE != 5;
        auto chain_pair = [&](int gi, auto with_loads) {
            constexpr bool LOADS = decltype(with_loads)::value;
            const int off0 = gi * g.S;
            const float* b0p = xs + base_full + off0;
            const float* b1p = b0p + g.S;
            const float* t0p = xs + base_tail + off0;
            const float* t1p = t0p + g.S;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const int o0 = out_lane + gi * out_step, o1 = o0 + out_step;
            if constexpr (ACC) {                                // start from the earlier passes' sum
                if (whole) {
                    acc0 = *reinterpret_cast<const f32x4*>(yc + o0);
                    if (kPair) acc1 = *reinterpret_cast<const f32x4*>(yc + o1);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (o0 + j >= lo && o0 + j < hi) acc0[j] = yc[o0 + j];
                        if (kPair && o1 + j >= lo && o1 + j < hi) acc1[j] = yc[o1 + j];
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
                if (kPair) w1[t % (kAhead + 1)] = t < 16 * FULL ? b1p[64 * (t / 16) + (t % 16)] : t1p[t - 16 * FULL];
            };
            if constexpr (AF_BANDED_ABL & 64) {                  // memory side only
                acc0[0] = a[0] + a[NK - 1];
                if (LOADS) {
#pragma unroll
                    for (int s2 = 0; s2 < kLoads; ++s2) issue_load(s2);
                }
            }
#pragma unroll
            for (int t = 0; t < kAhead && t < NK && !(AF_BANDED_ABL & 64); ++t) read_b(t);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < ((AF_BANDED_ABL & 64) ? 0 : NK); ++t) {
                if (t + kAhead < NK) read_b(t + kAhead);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], w0[t % (kAhead + 1)], acc0, 0, 0, 0);
                if (kPair) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], w1[t % (kAhead + 1)], acc1, 0, 0, 0);
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
            if constexpr (AF_BANDED_ABL & 2) {
                if (acc0[0] == 1.2345e-30f) yc[o0] = acc0[1];        // keeps the arithmetic alive, never taken in practice
            } else if constexpr (BS) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc0), yrs, (int)((mb32 + (uint32_t)o0) * 4u), 0, AF_Y_AUX);
            } else if (whole) {
                *reinterpret_cast<f32x4*>(yc + o0) = acc0;
                if (kPair) *reinterpret_cast<f32x4*>(yc + o1) = acc1;
            } else {                                            // clip edges, unaligned streams
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (o0 + j >= lo && o0 + j < hi) yc[o0 + j] = acc0[j];
                    if (kPair && o1 + j >= lo && o1 + j < hi) yc[o1 + j] = acc1[j];
                }
            }
        };
        chain_pair(0, std::true_type{});
        for (int gi = kPair ? 2 : 1; gi < g.gpw; gi += kPair ? 2 : 1) chain_pair(gi, std::false_type{});
    }
}

// ---- ratios with many block matrices: the matrices are streamed --------------------------------------------
// nb = L / gcd(16, L) > 16 (48 k -> 44.1 k: 147, 16 k -> 44.1 k: 441): a wave cannot own "its" matrix for the whole
// launch.  Blocks are numbered b = period * nb + p; a tile is the matrices [p0, p0 + P) of 16 C consecutive periods.
// Every period has one LDS segment holding x[period S + q(16 p0) + lo_off ...) for all P matrices of the range
// (their windows move on by about 16 M / L samples per matrix, so the range's windows overlap: each staged sample is
// used by several matrices; the W-sample halo between neighbouring ranges is re-read, mostly from L2).  Wave w takes
// the matrices p0 + w, p0 + w + 8, ...: it fetches the NK fragment registers of its NEXT matrix (a [nb][NK][64] table,
// L2-resident) while it runs the C chains of the current one (16 periods = the 16 MFMA columns), so the fragment
// traffic from L2 is 1 / C of what one chain per fetch would need.  k-slot mapping, fragment table and chain order
// are those of banded_kernel: a block's bits depend on (p, window) only, never on the tiling.
struct StreamGeom {
    const float* x;
    float* y;
    const float* afrag;              // [nb][NK][64]
    const int* qtab;                 // [nb + 1]: q(16 p), p = 0..nb (host-built; q(16 nb) = S)
    int64_t x_abs0, x_len, x_stride, y_stride, m0, n_out;
    int64_t pg0;                     // first period group (of 16 C periods) this launch touches
    int64_t n_pg, n_pr;              // period groups per clip, matrix ranges
    int64_t n_tiles;
    int lo_off;
    int nb, S, P, C, seg_len, seg_stride;
    int ob_off, ob_stride;           // deferred stores: the tile's outputs in LDS, [16 C periods][ob_stride] floats from ob_off
};

// wave-uniform table entry through the scalar cache (a vector load would count in vmcnt, and waiting for it would
// wait for every x request in flight)
__device__ __forceinline__ int sload_int(const int* base, int idx)
{
    typedef const __attribute__((address_space(4))) int* cptr;
    return reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(base))[__builtin_amdgcn_readfirstlane(idx)];
}

// AF_STREAM_ABL (diagnostic builds only, scripts/ablate/): parts of the streamed kernel compiled OUT, to see which
// resource its time belongs to: 1 x requests, 2 stores, 4 fragment requests (first tile's stay), 8 MFMAs (one VALU
// operation each instead), 16 staging writes, 32 barriers, 64 the whole chain arithmetic (memory side only), 128 stores go to the
// wrong place, 1 KB contiguous per instruction, 256 to the wrong place in 128-byte runs from two consecutive instructions,
// 512 to the wrong place in 128-byte runs inside one instruction, 1024 all fragment requests go to the first matrix's cache
// lines.  Results are garbage; bounds stay checked by the hardware.
#ifndef AF_STREAM_ABL
#define AF_STREAM_ABL 0
#endif
#ifndef AF_STREAM_OB
#define AF_STREAM_OB 1
#endif
// 1 (experiment, round 5): a workgroup takes a contiguous run of tiles instead of every gridDim-th one
#ifndef AF_STREAM_CONTIG
#define AF_STREAM_CONTIG 0
#endif
// largest matrix range of a tile (experiment: 16 = always small enough for the deferred stores)
#ifndef AF_STREAM_PMAX
#define AF_STREAM_PMAX 32
#endif
// OB (deferred stores; where two workgroups per CU still fit): a chain's lane holds 16 bytes of one block, and a store
// instruction straight from the accumulators writes 16 runs of 64 bytes, 16 nb floats apart -- measured on 48 -> 44.1 kHz
// (scripts/ablate/stream_groups.sh): the same bytes in 1 KB runs per instruction make the whole launch 10 % shorter, in
// 128-byte runs 4.6 %, and 64-byte pieces that are neighbours in TIME (two instructions) gain nothing.  So the chains
// put their blocks into an LDS image of the tile's outputs, [16 C periods][P blocks], and the workgroup sends it out at
// the top of the NEXT tile, between the two barriers that are there anyway: a period's row is P 64 bytes of consecutive
// output, one store instruction covers a whole row or more.
template <int NK, int C, bool OB>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void banded_stream_kernel(const StreamGeom g)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];
    constexpr int FULL = NK / 16, TAILQ = NK % 16;
    constexpr int kSegs = 16 * C;                                   // one segment per period of the tile
    float* ob = xs + g.ob_off;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g4 = lane >> 4;
    const uint32_t tiles_per_clip = (uint32_t)(g.n_pg * g.n_pr), n_tiles = (uint32_t)g.n_tiles;
    // x staging: a thread takes the sample pair (2 h, 2 h + 1), h = tid % 256, of every second segment (the upper
    // half of the workgroup the odd ones): 8-byte requests halve the vector-memory instructions of a tile, and the
    // two ds_write_b32 per pair are 2-way bank conflicts at most (free for stores)
    // The pair grid is anchored at an EVEN sample of the clip (a pair that straddled sample 0 would have a wrapped
    // byte offset and be dropped whole by the range check): when a tile's first sample is odd the pairs start one
    // sample early (sh = 1); segments keep one lead-in word for that sample.
    const int half = tid >> 8, pair = tid & 255;

    constexpr uint32_t kOob = 0xFFF00000u;
    constexpr int kLoads = kSegs / 2;
    float pf[kSegs];
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff;
    int sh = 0;                                                     // of the tile whose x is in flight (wave-uniform)
    // tile -> (clip, period group, matrix range); ranges vary fastest: neighbouring ranges share their halo in L2
    auto prep_tile = [&](uint32_t tile, bool live, int q_p0) {          // q_p0 = q(16 p0) of that tile
        const uint32_t clip = tile / tiles_per_clip, r = tile - clip * tiles_per_clip;
        const int64_t pg = g.pg0 + r / (uint32_t)g.n_pr;
        const int64_t base = pg * (int64_t)kSegs * g.S + q_p0 + g.lo_off - g.x_abs0;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x + (int64_t)clip * g.x_stride), 0,
                                                 (int)(uint32_t)(g.x_len * 4), 0x00020000);
        sh = (int)(base & 1);
        const bool first = 2 * pair < g.seg_len + sh;               // seg_len <= 511
        voff = live && first ? ((uint32_t)(base - sh) + (uint32_t)(2 * pair + half * g.S + opaque_zero())) * 4u : kOob;
    };
    auto issue_load = [&](int i) {                                  // segment 2 i + half
        if constexpr (AF_STREAM_ABL & 1) return;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 v = __builtin_bit_cast(
            f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(voff + (uint32_t)(2 * i * g.S) * 4u), 0, AF_X_AUX));
        // (everything in the VECTOR offset: the scalar offset operand is added behind the hardware's range check)
        pf[2 * i] = v[0];
        pf[2 * i + 1] = v[1];
    };

    float a[2][NK];                                                 // this matrix, the next one
    static_assert(NK % 4 == 0, "fragment quads");
    bool have_a = false;
    auto load_a = [&](int slot, int p) {
        if ((AF_STREAM_ABL & 4) && have_a && slot == 0) return;
        if ((AF_STREAM_ABL & 4) && slot == 1) return;
        have_a = true;
        // (AF_STREAM_ABL & 1024, diagnostic: every matrix's fragments read from the first matrix's lines -- the table stays in
        // the nearest cache, so what FETCH_SIZE loses against the product is the table's share of the fetched bytes)
        const f32x4* ap = reinterpret_cast<const f32x4*>(g.afrag) + ((size_t)((AF_STREAM_ABL & 1024) ? 0 : p) * (NK / 4)) * 64 + lane + opaque_zero();
#pragma unroll
        for (int t4 = 0; t4 < NK / 4; ++t4) {
            const f32x4 v = ap[t4 * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[slot][4 * t4 + j] = v[j];
        }
    };

    // first matrix range of a tile
    auto range_of = [&](uint32_t t, int& p0, int& p_end) {
        const uint32_t r = t % tiles_per_clip;
        p0 = (int)(r % (uint32_t)g.n_pr) * g.P;
        p_end = p0 + g.P < g.nb ? p0 + g.P : g.nb;
    };

    // q(16 p) of a tile's range start and of this wave's matrices in it, through the scalar cache -- requested a
    // whole tile ahead (an s_load consumed where it is issued costs the wave a few hundred cycles per tile)
    struct TileQ { int q0, q[4]; };
    auto load_q = [&](uint32_t t) {
        int p0, p_end;
        range_of(t, p0, p_end);
        TileQ v;
        v.q0 = sload_int(g.qtab, p0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = p0 + wave + 8 * i;
            v.q[i] = sload_int(g.qtab, p < g.nb ? p : g.nb);
        }
        return v;
    };

    // OB: the previous tile's outputs wait in LDS (wave-uniform description)
    bool prev = false;
    const float* prev_y = nullptr;                                  // its clip's row of y
    int64_t prev_m = 0;                                             // output index of its first block, minus m0
    int prev_np = 0;                                                // matrices in its range
    constexpr int kFlush = 4;                                       // 512 threads x 4 x 16 bytes: 32 KB, P <= 16
    auto flush = [&]() {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        // (every thread issues kFlush stores whatever the range holds: what lies outside goes to an offset the range
        // check drops, so the number of vector memory operations in front of the waits below does not depend on it)
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(prev_y), 0, (int)(uint32_t)(g.n_out * 4), 0x00020000);
        const uint32_t n4row = (uint32_t)prev_np * 4u;              // 16-byte pieces per period
        const uint32_t m_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)prev_m);
#pragma unroll
        for (int k = 0; k < kFlush; ++k) {
            const uint32_t f4 = (uint32_t)tid + 512u * k;
            const uint32_t row = f4 / n4row, c4 = f4 - row * n4row;
            const bool valid = row < (uint32_t)kSegs;
            const f32x4 v = *reinterpret_cast<const f32x4*>(ob + (valid ? row : 0u) * (uint32_t)g.ob_stride + 4u * c4);
            // 32 bits are enough: the tile grid reaches less than one tile past either end of the wanted outputs and
            // launch_stream bounds n_out, so nothing wraps INTO the range the hardware accepts
            const uint32_t off = (m_lo + row * (uint32_t)(16 * g.nb) + 4u * c4) * 4u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, (int)(valid ? off : kOob), 0, AF_Y_AUX);
        }
    };

    // AF_STREAM_CONTIG (experiment): a workgroup takes a CONTIGUOUS run of tiles (the matrix ranges of one period group vary
    // fastest over the tile index), so that the window halo between neighbouring ranges -- 27 % of what a range stages at
    // 48 -> 44.1 kHz, every bit of it an L2 miss: FETCH_SIZE 8.64 GB for 5.90 GB of x -- is re-read by the workgroup that
    // fetched it a tile ago.  Measured (profiles/r05_stream_traffic.txt): FETCH_SIZE does not move (4.23e6 against 4.22e6
    // units): an XCD's 64 resident workgroups stage and store 7.7 MB per tile time through a 4 MB L2, the halo is gone
    // before its second reader comes; 16 -> 44.1 kHz is 6 % slower.  The default stays round-robin.
#if AF_STREAM_CONTIG
    const uint32_t tile_end = (uint32_t)(((uint64_t)n_tiles * (blockIdx.x + 1)) / gridDim.x);
    uint32_t tile = (uint32_t)(((uint64_t)n_tiles * blockIdx.x) / gridDim.x);
    constexpr uint32_t kTileStepGrid = 0;
#else
    const uint32_t tile_end = n_tiles;
    uint32_t tile = blockIdx.x;
    constexpr uint32_t kTileStepGrid = 1;
#endif
    TileQ tq{};
    if (tile < tile_end) {
        tq = load_q(tile);
        prep_tile(tile, true, tq.q0);
#pragma unroll
        for (int i = 0; i < kLoads; ++i) issue_load(i);
    }
    for (; tile < tile_end;) {
        const uint32_t next_tile = tile + (kTileStepGrid ? gridDim.x : 1u);
        const uint32_t clip = tile / tiles_per_clip, r = tile - clip * tiles_per_clip;
        const int64_t pg = g.pg0 + r / (uint32_t)g.n_pr;
        int p0, p_end;
        range_of(tile, p0, p_end);
        const bool live = next_tile < tile_end;
        const TileQ cq = tq;                                        // this tile's; the next tile's are requested now
        if (live) tq = load_q(next_tile);
        // matrices p0 + wave + 8 i, i < n_m.  Both fragment slots are free here: the first matrix is requested now and
        // lands under the staging below.  (The first version requested it under the previous tile's last chain, into
        // whichever slot was free, and moved it to slot 0 when that was slot 1: with three matrices per wave -- the
        // 48 <-> 44.1 kHz case -- every tile ended on a full L2 round trip in front of 20 v_mov.)
        // A wave without a matrix in this range (the last, shorter range of a period only) still runs ONE chain, on the
        // range's last matrix with its stores switched off: the x requests of the next tile then sit in the same place
        // of the instruction stream for every wave (as a separate branch their staging registers were "possibly in
        // flight" at the join in front of the first chain, and that chain began with s_waitcnt vmcnt(0) in every tile).
        const bool idle = p0 + wave >= p_end;
        const int n_m = idle ? 1 : (p_end - p0 - wave + 7) / 8;
        const int pw = idle ? p_end - 1 : p0 + wave;
        load_a(0, pw);
        if constexpr (!(AF_STREAM_ABL & 32)) __syncthreads();       // readers of the previous tile done
        if constexpr (OB) {
            if (prev) flush();                                      // (its chains are all behind the barrier above)
        }
        if (2 * pair < g.seg_len + sh && !(AF_STREAM_ABL & 16)) {
            float* dst = xs + 1 - sh + 2 * pair + half * g.seg_stride + opaque_zero();
#pragma unroll
            for (int i = 0; i < kLoads; ++i) {
                dst[2 * i * g.seg_stride] = pf[2 * i];
                dst[2 * i * g.seg_stride + 1] = pf[2 * i + 1];
            }
        }
        if constexpr (!(AF_STREAM_ABL & 32)) __syncthreads();
        prep_tile(live ? next_tile : tile, live, tq.q0);
        float* yc = g.y + (int64_t)clip * g.y_stride - g.m0;        // where absolute output 0 of the clip would be
        const int q0 = cq.q0;
        // one chain pair: matrix p (fragments in a[SLOT]) times the 16 periods of column set 0 and of column set 1, two
        // independent accumulators fed by the same fragment registers; a wave's first pair also carries the next
        // tile's x requests between its steps
        static_assert(C == 2, "chain pairs");
        auto chain = [&](int p, int qp, auto slot_c, auto with_loads) {   // qp = q(16 p)
            constexpr bool LOADS = decltype(with_loads)::value;
            constexpr int SLOT = decltype(slot_c)::value;
            const int qrel = qp - q0;                               // window start of matrix p inside a segment
            const float* bfull = xs + 1 + col * g.seg_stride + qrel + 16 * g4;       // (+ 1: the lead-in word)
            const float* btail = xs + 1 + col * g.seg_stride + qrel + TAILQ * g4 + 64 * FULL;
            const int set1 = 16 * g.seg_stride;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            constexpr int kAhead = 3;
            float w0[kAhead + 1], w1[kAhead + 1];
            auto read_b = [&](int t) {
                const float* src = t < 16 * FULL ? bfull + 64 * (t / 16) + (t % 16) : btail + (t - 16 * FULL);
                w0[t % (kAhead + 1)] = src[0];
                w1[t % (kAhead + 1)] = src[set1];
            };
            if constexpr (AF_STREAM_ABL & 64) {                      // memory side only: requests, no LDS reads, no MFMA
                acc0[0] = a[SLOT][0] + a[SLOT][NK - 1];
                acc1[0] = a[SLOT][1];
                if (LOADS) {
#pragma unroll
                    for (int s2 = 0; s2 < kLoads; ++s2) issue_load(s2);
                }
            }
#pragma unroll
            for (int t = 0; t < kAhead && t < NK && !(AF_STREAM_ABL & 64); ++t) read_b(t);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < ((AF_STREAM_ABL & 64) ? 0 : NK); ++t) {
                if (t + kAhead < NK) read_b(t + kAhead);
                if constexpr (AF_STREAM_ABL & 8) {
                    acc0[t & 3] = __builtin_fmaf(a[SLOT][t], w0[t % (kAhead + 1)], acc0[t & 3]);
                    acc1[t & 3] = __builtin_fmaf(a[SLOT][t], w1[t % (kAhead + 1)], acc1[t & 3]);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[SLOT][t], w0[t % (kAhead + 1)], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[SLOT][t], w1[t % (kAhead + 1)], acc1, 0, 0, 0);
                }
                if (LOADS) {
                    constexpr int kSpan = (NK + 1) / 2;
                    if (t < kSpan) {
#pragma unroll
                        for (int s2 = (kLoads * t) / kSpan; s2 < (kLoads * (t + 1)) / kSpan; ++s2) issue_load(s2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // block b = period * nb + p of column `col`: rows 4 g4 .. 4 g4 + 3
            if constexpr (OB) {
                if (!idle) {
                    float* o = ob + col * g.ob_stride + (p - p0) * 16 + 4 * g4;
                    *reinterpret_cast<f32x4*>(o) = acc0;
                    *reinterpret_cast<f32x4*>(o + 16 * g.ob_stride) = acc1;
                }
                return;
            }
#pragma unroll
            for (int cs = 0; cs < 2; ++cs) {
                const f32x4 acc = cs ? acc1 : acc0;
                const int64_t period = pg * kSegs + cs * 16 + col;
                int64_t m = (idle || (AF_STREAM_ABL & 2)) ? -8 : 16 * (period * g.nb + p) + 4 * g4;     // (idle: in front of every m0)
                if constexpr (AF_STREAM_ABL & 256)                  // (diagnostic: the two column sets' chunks side by side: 128 B runs)
                    m = 16 * ((pg * kSegs + col) * (int64_t)g.nb + (p & ~1) + cs) + 4 * g4 + ((p & 1) ? 16 * (int64_t)g.nb * 16 : 0);
                if constexpr (AF_STREAM_ABL & 512)                  // (diagnostic: wrong place, 128-byte runs inside ONE instruction)
                    m = 16 * ((pg * kSegs + cs * 16 + (col & ~1) + (p & 1)) * (int64_t)g.nb + (p & ~1) + (col & 1)) + 4 * g4;
                if constexpr (AF_STREAM_ABL & 128)                  // (diagnostic: wrong place, 1 KB contiguous per instruction)
                    m = 16 * ((pg * kSegs + cs * 16) * (int64_t)g.nb + (int64_t)p * 16) + 4 * lane;
                float* o = yc + m;
                if (m >= g.m0 && m + 4 <= g.m0 + g.n_out && ((reinterpret_cast<uintptr_t>(o) >> 2) & 3) == 0) {
                    *reinterpret_cast<f32x4*>(o) = acc;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (m + j >= g.m0 && m + j < g.m0 + g.n_out) o[j] = acc[j];
                }
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        // fragment slots alternate at compile time (a runtime slot index would cost a select per MFMA operand)
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                // P <= 32: at most four matrices per wave
            if (i == 0 || i < n_m) {                                 // (n_m >= 1: the first chain is unconditional)
                const int p = pw + 8 * i;
                if (i + 1 < n_m) load_a((i & 1) ^ 1, p + 8);
                if (i == 0) chain(p, idle ? sload_int(g.qtab, p) : cq.q[0], S0{}, std::true_type{});
                else if ((i & 1) == 0) chain(p, cq.q[i], S0{}, std::false_type{});
                else chain(p, cq.q[i], S1{}, std::false_type{});
            }
        }
        if constexpr (OB) {
            prev = true;
            prev_y = g.y + (int64_t)clip * g.y_stride;
            prev_m = 16 * ((pg * kSegs) * (int64_t)g.nb + p0) - g.m0;
            prev_np = p_end - p0;
        }
        tile = next_tile;
    }
    if constexpr (OB) {
        if (prev) {
            __syncthreads();
            flush();
        }
    }
}

// ---- resampler + FIR of BASELINE config 4 in one launch -----------------------------------------
// Phase A is the banded resampler above (one block matrix, 8 waves) writing its 256 G outputs of a
// tile into LDS instead of HBM, already cut into the 16 column segments of phase B; phase B is the
// banded FIR over those segments.  A FIR output needs K - 1 resampled samples of history, so
// consecutive tiles overlap by H2 = 16 ceil((K - 1) / 16) resampled samples (recomputed, 1 / G of the
// phase-A work) and a tile yields P2 = G - 1 FIR block periods per column.  Block alignment, k-slot
// mapping and chain order are those of the stand-alone kernels, so with the same NK the output is
// theirs bit for bit.
struct FusedGeom {
    const void* x;                   // [clip][frame][channels], f32 or i16 (FMT)
    float* z;                        // [clip][n_out]: FIR outputs m0 .. m0 + n_out of every clip
    const float* afrag_a;            // resampler fragments [NKA][64]
    const float* afrag_b;            // FIR fragments [NKB][64]
    const FusedTile* tiles;          // ragged batch: one record per tile; null: uniform clips (arithmetic below)
    int64_t x_abs0;                  // absolute stream index of x[0] of every clip (even in the pair formats)
    int64_t x_len, x_stride, z_stride;
    int64_t n_res;                   // resampled samples that exist: [0, n_res); the FIR sees zeros before 0
    int64_t m0, n_out;               // FIR outputs wanted (absolute indices), tiles are anchored at output 0
    int64_t tile0, tiles_per_clip, n_tiles;
    int lo_off;                      // resampler: c - (T - 1)
    int S, G, gpw, seg_len, seg_stride;
    int shift;                       // pair formats: frames in front of a segment so that frame pairs load aligned
    int H2, pad, P2, seg_len_b, seg_stride_b, ys_off;
    float inv_colz;                  // 1 / (16 P2)
    // resfir10_kernel (ratios with ten block matrices) only:
    uint32_t L, M;                   // q(16 p) = floor(16 p M / L) per wave
    int K;                           // FIR taps: the K - 1 resampled samples in front of a tile are carried in LDS
    int carry_off;                   // floats from the start of LDS: two carry buffers of kCarry floats
};

// What a tile needs to know about its clip (wave-uniform: SGPRs).
struct TileCtx {
    const char* xp;                  // the clip's first byte
    float* zp;                       // where FIR output 0 of the clip would be stored (z row - m0)
    int64_t n_res, t_abs, m_lo, m_hi;
    uint32_t x_bytes;
};

// One pair of MFMA chains over two block periods; `issue(t)` may put global loads between the steps.
template <int NK, class Issue>
__device__ __forceinline__ void chain2(const float (&a)[NK], const float* b0p, const float* b1p, const float* t0p,
                                       const float* t1p, f32x4& acc0, f32x4& acc1, Issue&& issue)
{
    constexpr int FULL = NK / 16;
    constexpr int kAhead = 3;
    float w0[kAhead + 1], w1[kAhead + 1];
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
        issue(t);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// FMT: 0 = mono f32 (a thread stages samples tid and tid + 512 of every column segment, one dword load each);
// pair formats (a thread stages the frame pair (2 tid, 2 tid + 1) of every segment with ONE load; the segment starts
// `shift` frames early so that a pair never straddles the clip start and the buffer's range check only ever zeroes
// whole pairs' dwords): 1 = stereo f32 (16 bytes: L0 R0 L1 R1), 2 = stereo i16 (8 bytes), 3 = mono i16 (4 bytes).
// Stereo is downmixed on its way into LDS ((L + R) * 0.5, the to_mono of capture.rs:30-42), i16 samples are scaled
// by 1 / 32768 first (docs/OPS.md "PCM formats": the arithmetic of the feature kernels' Stage<>).
template <int FMT>
struct FusedFmt {
    static constexpr bool kPair = FMT != 0;
    static constexpr int kBytesPerFrame = FMT == 0 ? 4 : FMT == 1 ? 8 : FMT == 2 ? 4 : 2;
    static constexpr int kRegsPerCol = FMT == 0 ? kPfSlots : FMT == 1 ? 4 : FMT == 2 ? 2 : 1;
};

template <int NKA, int NKB, int FMT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void resfir_kernel(const FusedGeom g)
{
    using F = FusedFmt<FMT>;
    constexpr bool PAIR = F::kPair;
    constexpr int BPF = F::kBytesPerFrame;
    extern __shared__ __attribute__((aligned(16))) float xs[];
    float* ys = xs + g.ys_off;
    constexpr int FULLA = NKA / 16, TAILA = NKA % 16, FULLB = NKB / 16, TAILB = NKB % 16;
    constexpr int nthreads = 512;
    constexpr int kGpw = 2;                                         // G = 16: one chain pair per wave and phase
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g4 = lane >> 4;

    const int wave_off_a = wave * kGpw * g.S;
    const int base_full_a = col * g.seg_stride + 16 * g4 + wave_off_a + g.shift;
    const int base_tail_a = col * g.seg_stride + TAILA * g4 + 64 * FULLA + wave_off_a + g.shift;
    const int wave_off_b = wave * kGpw * 16;
    const int base_full_b = col * g.seg_stride_b + 16 * g4 + wave_off_b;
    const int base_tail_b = col * g.seg_stride_b + TAILB * g4 + 64 * FULLB + wave_off_b;
    const int y_lane = 16 * (col * g.G + wave * kGpw) + 4 * g4;     // tile-relative resampled index of acc0[0]
    const int z_lane = 16 * (col * g.P2 + wave * kGpw) + 4 * g4;    // tile-relative FIR output of acc0[0]
    const int colz = 16 * g.P2, z_per_tile = 16 * colz, y_per_tile = 256 * g.G;
    const int col_span = g.G * g.S;
    const bool first = PAIR ? 2 * tid < g.seg_len + g.shift : tid < g.seg_len;
    const bool second = !PAIR && tid + nthreads < g.seg_len;
    const uint32_t tpc = (uint32_t)g.tiles_per_clip, n_tiles = (uint32_t)g.n_tiles;

    // The clip of a tile: arithmetic for uniform batches, one 32-byte record (scalar load) for ragged ones.
    auto tile_ctx = [&](uint32_t tile) {
        TileCtx c;
        if (g.tiles) {
            typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
            typedef const __attribute__((address_space(4))) u32x8* cptr;
            const u32x8 r = reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(g.tiles))[__builtin_amdgcn_readfirstlane(tile)];
            const int64_t x_off = (int64_t)(((unsigned long long)r[1] << 32) | r[0]);
            const int64_t z_off = (int64_t)(((unsigned long long)r[3] << 32) | r[2]);
            c.xp = static_cast<const char*>(g.x) + x_off * BPF;
            c.zp = g.z + z_off;
            c.x_bytes = r[4] * (uint32_t)BPF;
            c.n_res = (int64_t)r[5];
            c.t_abs = (int64_t)r[6];
            c.m_lo = 0;
            c.m_hi = c.n_res;
        } else {
            const uint32_t clip = tile / tpc;
            c.xp = static_cast<const char*>(g.x) + (int64_t)clip * g.x_stride * BPF;
            c.zp = g.z + (int64_t)clip * g.z_stride - g.m0;
            c.x_bytes = (uint32_t)(g.x_len * BPF);
            c.n_res = g.n_res;
            c.t_abs = g.tile0 + (tile - clip * tpc);
            c.m_lo = g.m0;
            c.m_hi = g.m0 + g.n_out < g.n_res ? g.m0 + g.n_out : g.n_res;
        }
        return c;
    };

    // Register plan (128 per lane, 4 waves per SIMD): a phase only holds its own fragments -- they are
    // fetched again (L2) per tile, the resampler's before the first barrier, the FIR's while the
    // resampled samples go to LDS.  The next tile's x is requested in two halves: slot 0 (mono f32: all
    // of it) in the first half of phase A, slot 1 step by step through phase B while the FIR fragments
    // are dying, so at most 32 staging registers are live next to a full set of fragments.
    float a1[NKA], a2[NKB];
    auto load_a1 = [&](int oz) {
#pragma unroll
        for (int t = 0; t < NKA; ++t) a1[t] = g.afrag_a[t * 64 + lane + oz];
    };
    auto load_a2 = [&](int oz) {
#pragma unroll
        for (int t = 0; t < NKB; ++t) a2[t] = g.afrag_b[t * 64 + lane + oz];
    };

    constexpr uint32_t kOob = 0xFFF00000u;
    constexpr int kLoads = PAIR ? kCols : kCols * kPfSlots;          // load instructions per thread and tile
    constexpr int kLoadsA = FMT == 1 ? kCols / 2 : kLoads;           // requested during phase A
    constexpr int kRegs = kCols * F::kRegsPerCol;
    float pf[kRegs];
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[kPfSlots];
    uint32_t x_frames = 0;           // FMT 3: frames of the clip whose tile is staged next (see staged_pair)
    auto prep_tile = [&](uint32_t tile, bool live) {
        const TileCtx c = tile_ctx(tile);
        // x index of the first staged sample of the tile: q(y0) + lo_off - shift, y0 = t z_per_tile - H2,
        // relative to the buffer's first frame
        const int64_t base = (c.t_abs * (int64_t)(16 * g.P2) - g.H2 / 16) * g.S + g.lo_off - g.shift - g.x_abs0;
        // mono i16 is staged as one dword per frame pair and the range check works per dword: for an odd frame count
        // the last dword (frames n - 1 and n) would be read or dropped whole.  The range is rounded up to whole dwords
        // (the extra 2 bytes share an aligned dword with the clip's last sample) and staged_pair zeroes the half that
        // lies behind the clip.
        if constexpr (FMT == 3) x_frames = c.x_bytes >> 1;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(c.xp), 0, (int)(FMT == 3 ? (c.x_bytes + 2u) & ~3u : c.x_bytes),
                                                 0x00020000);
        if constexpr (PAIR) {
            const uint32_t v0 = ((uint32_t)base + 2u * (uint32_t)(tid + opaque_zero())) * (uint32_t)BPF;
            voff[0] = live && first ? v0 : kOob;
        } else {
            const uint32_t v0 = ((uint32_t)base + (uint32_t)(tid + opaque_zero())) * 4u;
            voff[0] = live && first ? v0 : kOob;
            voff[1] = live && second ? v0 + (uint32_t)nthreads * 4u : kOob;
        }
    };
    auto issue_load = [&](int s) {
        if constexpr (FMT == 0) {                                   // s = h * 16 + cc
            const int cc = s % kCols, h = s / kCols;
            pf[cc * kPfSlots + h] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(voff[h] + (uint32_t)(cc * col_span) * 4u), 0, AF_X_AUX));
        } else if constexpr (FMT == 1) {                            // s = cc
            // (cast the whole vector: hipcc 7.2 reads element 0 for __builtin_bit_cast(float, v[i]))
            const f32x4 v = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(voff[0] + (uint32_t)(s * col_span) * 8u), 0, AF_X_AUX));
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[s * 4 + i] = v[i];
        } else if constexpr (FMT == 2) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 v = __builtin_bit_cast(
                f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(voff[0] + (uint32_t)(s * col_span) * 4u), 0, AF_X_AUX));
            pf[s * 2] = v[0];
            pf[s * 2 + 1] = v[1];
        } else {
            pf[s] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(voff[0] + (uint32_t)(s * col_span) * 2u), 0, AF_X_AUX));
        }
    };
    // the frame pair of column cc as two mono samples
    auto staged_pair = [&](int cc, float& m0, float& m1) {
        auto lo16 = [](float w) { return (float)(int16_t)(__float_as_uint(w) & 0xffffu) * (1.0f / 32768.0f); };
        auto hi16 = [](float w) { return (float)(int16_t)(__float_as_uint(w) >> 16) * (1.0f / 32768.0f); };
        if constexpr (FMT == 1) {
            m0 = (pf[cc * 4] + pf[cc * 4 + 1]) * 0.5f;
            m1 = (pf[cc * 4 + 2] + pf[cc * 4 + 3]) * 0.5f;
        } else if constexpr (FMT == 2) {
            m0 = (lo16(pf[cc * 2]) + hi16(pf[cc * 2])) * 0.5f;
            m1 = (lo16(pf[cc * 2 + 1]) + hi16(pf[cc * 2 + 1])) * 0.5f;
        } else {
            m0 = lo16(pf[cc]);
            // byte offset of the pair's second sample, with the wrap-around of the load's own offset arithmetic (a
            // column of a segment that starts before the clip can lie inside it; an offset before the clip is huge)
            const uint32_t b1 = voff[0] + (uint32_t)(cc * col_span) * 2u + 2u;
            m1 = b1 < 2u * x_frames ? hi16(pf[cc]) : 0.0f;
        }
    };

    uint32_t tile = blockIdx.x;
    if (tile < n_tiles) {
        prep_tile(tile, true);
#pragma unroll
        for (int s2 = 0; s2 < kLoads; ++s2) issue_load(s2);
    }
    for (; tile < n_tiles; tile += gridDim.x) {
        // No barrier here: xs was last read in phase A and ys in phase B of the previous tile; a wave
        // writes xs before, and ys after, the next barrier, which every wave reaches after its phase B.
        {
            if constexpr (PAIR) {
                float* dst = xs + 2 * (tid + opaque_zero());
                if (first) {
#pragma unroll
                    for (int cc = 0; cc < kCols; ++cc) {
                        float m0, m1;
                        staged_pair(cc, m0, m1);
                        dst[cc * g.seg_stride] = m0;
                        dst[cc * g.seg_stride + 1] = m1;
                    }
                }
            } else {
                float* dst = xs + tid + opaque_zero();
                if (first) {
#pragma unroll
                    for (int cc = 0; cc < kCols; ++cc) dst[cc * g.seg_stride] = pf[cc * kPfSlots];
                }
                if (second) {
#pragma unroll
                    for (int cc = 0; cc < kCols; ++cc) dst[cc * g.seg_stride + nthreads] = pf[cc * kPfSlots + 1];
                }
            }
        }
        load_a1(opaque_zero());
        __syncthreads();
        const TileCtx tc = tile_ctx(tile);
        {
            const bool live = tile + gridDim.x < n_tiles;
            prep_tile(live ? tile + gridDim.x : tile, live);
        }

        // ---- phase A: resampled samples [y0, y0 + 256 G) into the FIR's column segments ----
        const int64_t z0 = tc.t_abs * z_per_tile;                    // first FIR output of the tile
        {
            const int64_t y0 = z0 - g.H2;
            // resampled samples outside [0, n_res) are the FIR's zero initial state / never used
            const int y_lo = y0 >= 0 ? 0 : (int)-y0;
            const int64_t y_left = tc.n_res - y0;
            const int y_hi = y_left > y_per_tile ? y_per_tile : (int)y_left;
            const float* b0p = xs + base_full_a;
            const float* t0p = xs + base_tail_a;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            chain2<NKA>(a1, b0p, b0p + g.S, t0p, t0p + g.S, acc0, acc1, [&](int t) {
                constexpr int kSpan = (NKA + 1) / 2;
                if (t < kSpan) {
#pragma unroll
                    for (int s2 = (kLoadsA * t) / kSpan; s2 < (kLoadsA * (t + 1)) / kSpan; ++s2) issue_load(s2);
                }
            });
            load_a2(opaque_zero());
            // branch-free: a sample that belongs to no segment (or to only one) goes to this lane's
            // spare word behind the segments
            const int spare = kCols * g.seg_stride_b + lane;
            auto put_y = [&](int yi, float v) {
                v = (yi >= y_lo && yi < y_hi) ? v : 0.0f;
                const int e = yi - g.pad;
                const int c = (int)(((float)e + 0.5f) * g.inv_colz);    // e / colz for e >= 0
                const int pos = e - c * colz;
                const int at = c * g.seg_stride_b + pos;
                ys[(e >= 0 && c < kCols) ? at : spare] = v;
                ys[(e >= colz && c <= kCols && pos + colz < g.seg_len_b) ? at - g.seg_stride_b + colz : spare] = v;
            };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                put_y(y_lane + j, acc0[j]);
                put_y(y_lane + 16 + j, acc1[j]);
            }
        }
        __syncthreads();

        // ---- phase B: the FIR over the segments, P2 valid block periods per column ----
        {
            float* zc = tc.zp + z0;
            // outputs [z_lo, z_hi) of the tile are wanted (absolute m_lo <= z0 + o < m_hi)
            const int64_t lo_left = tc.m_lo - z0, hi_left = tc.m_hi - z0;
            const int z_lo = lo_left <= 0 ? 0 : lo_left > z_per_tile ? z_per_tile : (int)lo_left;
            const int z_hi = hi_left <= 0 ? 0 : hi_left > z_per_tile ? z_per_tile : (int)hi_left;
            const bool whole = z_lo == 0 && z_hi == z_per_tile && ((reinterpret_cast<uintptr_t>(zc) >> 2) & 3) == 0;
            const float* b0p = ys + base_full_b;
            const float* t0p = ys + base_tail_b;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            chain2<NKB>(a2, b0p, b0p + 16, t0p, t0p + 16, acc0, acc1, [&](int t) {
                // one request per step from the start: a fragment register dies per step, and the
                // earlier a load goes out the more of the phase covers its latency
                constexpr int kFrom = 2, kSpan = NKB - 4;
                if (kLoadsA < kLoads && t >= kFrom && t < kFrom + kSpan) {
#pragma unroll
                    for (int s2 = kLoadsA + ((kLoads - kLoadsA) * (t - kFrom)) / kSpan;
                         s2 < kLoadsA + ((kLoads - kLoadsA) * (t - kFrom + 1)) / kSpan; ++s2)
                        issue_load(s2);
                }
            });
            const int pr0 = wave * kGpw;                             // block period of acc0 (wave-uniform)
            const int o0 = z_lane, o1 = o0 + 16;
            if (whole) {
                if (pr0 < g.P2) *reinterpret_cast<f32x4*>(zc + o0) = acc0;
                if (pr0 + 1 < g.P2) *reinterpret_cast<f32x4*>(zc + o1) = acc1;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (pr0 < g.P2 && o0 + j >= z_lo && o0 + j < z_hi) zc[o0 + j] = acc0[j];
                    if (pr0 + 1 < g.P2 && o1 + j >= z_lo && o1 + j < z_hi) zc[o1 + j] = acc1[j];
                }
            }
        }
    }
}


// ---- resampler + FIR in one launch for ratios with TEN block matrices (44.1 k -> 16 k: L / M = 160 / 441) -------------
// The everyday device rate (capture.rs:60-80 takes what the device offers; BASELINE config 3's ratio).  Ten waves, wave p
// owns block matrix p for the whole launch (its fragments and the FIR's stay in registers: one workgroup per CU leaves
// 168 registers per lane); a tile is 16 columns x 2 block periods x 10 matrices x 16 = 5 120 resampled samples (phase A,
// banded_kernel's ten-wave shape) and the SAME 5 120 FIR outputs (phase B: 20 block periods per column, one chain pair per
// wave) -- nothing is recomputed: the K - 1 resampled samples in front of a tile are the previous tile's last ones and stay
// in LDS.  For that a workgroup takes a CONTIGUOUS range of tiles; where a range starts inside a clip, phase A of the tile
// in front of it runs first ("warm": no FIR, no output), once per range.  Block alignment, k-slot mapping and chain
// order are those of the stand-alone kernels, tiles are anchored at absolute output 0 (5 120 = 32 rounds of the ten
// matrices), so the bits do not depend on how a batch or a stream was cut.
constexpr int kCarry = 128;           // K - 1 <= 128 resampled samples carried between consecutive tiles

template <int NKA, int NKB, int FMT>
__global__ __launch_bounds__(640) __attribute__((amdgpu_waves_per_eu(3, 3))) void resfir10_kernel(const FusedGeom g)
{
    using F = FusedFmt<FMT>;
    constexpr bool PAIR = F::kPair;
    constexpr int BPF = F::kBytesPerFrame;
    extern __shared__ __attribute__((aligned(16))) float xs[];
    float* ys = xs + g.ys_off;
    float* cw = xs + g.carry_off;
    constexpr int FULLA = NKA / 16, TAILA = NKA % 16, FULLB = NKB / 16, TAILB = NKB % 16;
    constexpr int nthreads = 640, NB = 10, kG = 2, P2 = 20, colz = 16 * P2, y_per_tile = 256 * P2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // = block matrix p
    const int qp = (int)(((int64_t)16 * wave * g.M) / g.L);          // q(16 p)

    const int wave_off_b = wave * 2 * 16;
    const int col_span = kG * g.S;
    const int km1 = g.K - 1;
    const bool first = PAIR ? 2 * tid < g.seg_len + g.shift : tid < g.seg_len;
    const bool second = !PAIR && tid + nthreads < g.seg_len;
    const uint32_t tpc = (uint32_t)g.tiles_per_clip, n_tiles = (uint32_t)g.n_tiles;

    // this workgroup's contiguous range of tiles
    const uint32_t r0 = (uint32_t)(((uint64_t)n_tiles * blockIdx.x) / gridDim.x);
    const uint32_t r1 = (uint32_t)(((uint64_t)n_tiles * (blockIdx.x + 1)) / gridDim.x);
    if (r0 >= r1) return;

    // (tile, warm): warm = phase A of the tile IN FRONT of `tile` (same clip), for its last K - 1 samples
    auto tile_ctx = [&](uint32_t tile, bool warm) {
        TileCtx c;
        if (g.tiles) {
            typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
            typedef const __attribute__((address_space(4))) u32x8* cptr;
            const u32x8 r = reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(g.tiles))[__builtin_amdgcn_readfirstlane(tile)];
            const int64_t x_off = (int64_t)(((unsigned long long)r[1] << 32) | r[0]);
            const int64_t z_off = (int64_t)(((unsigned long long)r[3] << 32) | r[2]);
            c.xp = static_cast<const char*>(g.x) + x_off * BPF;
            c.zp = g.z + z_off;
            c.x_bytes = r[4] * (uint32_t)BPF;
            c.n_res = (int64_t)r[5];
            c.t_abs = (int64_t)r[6];
            c.m_lo = 0;
            c.m_hi = c.n_res;
        } else {
            const uint32_t clip = tile / tpc;
            c.xp = static_cast<const char*>(g.x) + (int64_t)clip * g.x_stride * BPF;
            c.zp = g.z + (int64_t)clip * g.z_stride - g.m0;
            c.x_bytes = (uint32_t)(g.x_len * BPF);
            c.n_res = g.n_res;
            c.t_abs = g.tile0 + (tile - clip * tpc);
            c.m_lo = g.m0;
            c.m_hi = g.m0 + g.n_out < g.n_res ? g.m0 + g.n_out : g.n_res;
        }
        if (warm) c.t_abs -= 1;
        return c;
    };
    // does `tile` continue the tile before it (same clip, next t)?  Otherwise it starts a clip (or a stream window).
    auto continues = [&](uint32_t tile) {
        if (g.tiles) {
            typedef const __attribute__((address_space(4))) int* cptr;
            return reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(g.tiles))[8 * (size_t)__builtin_amdgcn_readfirstlane(tile) + 6] != 0;
        }
        return tile % tpc != 0;
    };
    auto needs_warm = [&](uint32_t tile, bool prev_done) {
        // the K - 1 samples in front of the tile: zeros at a clip's start, the previous tile's if this workgroup has just
        // computed it, otherwise a warm pass
        if (prev_done && continues(tile)) return false;
        return tile_ctx(tile, false).t_abs > 0;
    };

    float a1[NKA], a2[NKB];
#pragma unroll
    for (int t = 0; t < NKA; ++t) a1[t] = g.afrag_a[((size_t)wave * NKA + t) * 64 + lane];
#pragma unroll
    for (int t = 0; t < NKB; ++t) a2[t] = g.afrag_b[t * 64 + lane];
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0), visible to the compiler (see wire_kernel)
#pragma unroll
    for (int t = 0; t < NKA; ++t) asm volatile("" : "+v"(a1[t]));
#pragma unroll
    for (int t = 0; t < NKB; ++t) asm volatile("" : "+v"(a2[t]));

    constexpr uint32_t kOob = 0xFFF00000u;
    constexpr int kLoads = PAIR ? kCols : kCols * kPfSlots;
    constexpr int kRegs = kCols * F::kRegsPerCol;
    float pf[kRegs];
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[kPfSlots];
    uint32_t x_frames = 0;
    auto prep_tile = [&](uint32_t tile, bool warm, bool live) {
        const TileCtx c = tile_ctx(tile, warm);
        // x index of the first staged sample of the tile: 32 rounds of the ten matrices = 32 S input samples per tile
        const int64_t base = c.t_abs * (int64_t)(16 * kG) * g.S + g.lo_off - g.shift - g.x_abs0;
        if constexpr (FMT == 3) x_frames = c.x_bytes >> 1;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(c.xp), 0, (int)(FMT == 3 ? (c.x_bytes + 2u) & ~3u : c.x_bytes),
                                                 0x00020000);
        if constexpr (PAIR) {
            const uint32_t v0 = ((uint32_t)base + 2u * (uint32_t)(tid + opaque_zero())) * (uint32_t)BPF;
            voff[0] = live && first ? v0 : kOob;
        } else {
            const uint32_t v0 = ((uint32_t)base + (uint32_t)(tid + opaque_zero())) * 4u;
            voff[0] = live && first ? v0 : kOob;
            voff[1] = live && second ? v0 + (uint32_t)nthreads * 4u : kOob;
        }
    };
    auto issue_load = [&](int s) {
        if constexpr (FMT == 0) {                                   // s = h * 16 + cc
            const int cc = s % kCols, h = s / kCols;
            pf[cc * kPfSlots + h] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(voff[h] + (uint32_t)(cc * col_span) * 4u), 0, AF_X_AUX));
        } else if constexpr (FMT == 1) {                            // s = cc
            const f32x4 v = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(voff[0] + (uint32_t)(s * col_span) * 8u), 0, AF_X_AUX));
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[s * 4 + i] = v[i];
        } else if constexpr (FMT == 2) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 v = __builtin_bit_cast(
                f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(voff[0] + (uint32_t)(s * col_span) * 4u), 0, AF_X_AUX));
            pf[s * 2] = v[0];
            pf[s * 2 + 1] = v[1];
        } else {
            pf[s] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(voff[0] + (uint32_t)(s * col_span) * 2u), 0, AF_X_AUX));
        }
    };
    auto staged_pair = [&](int cc, float& m0, float& m1) {
        auto lo16 = [](float w) { return (float)(int16_t)(__float_as_uint(w) & 0xffffu) * (1.0f / 32768.0f); };
        auto hi16 = [](float w) { return (float)(int16_t)(__float_as_uint(w) >> 16) * (1.0f / 32768.0f); };
        if constexpr (FMT == 1) {
            m0 = (pf[cc * 4] + pf[cc * 4 + 1]) * 0.5f;
            m1 = (pf[cc * 4 + 2] + pf[cc * 4 + 3]) * 0.5f;
        } else if constexpr (FMT == 2) {
            m0 = (lo16(pf[cc * 2]) + hi16(pf[cc * 2])) * 0.5f;
            m1 = (lo16(pf[cc * 2 + 1]) + hi16(pf[cc * 2 + 1])) * 0.5f;
        } else {
            m0 = lo16(pf[cc]);
            const uint32_t b1 = voff[0] + (uint32_t)(cc * col_span) * 2u + 2u;
            m1 = b1 < 2u * x_frames ? hi16(pf[cc]) : 0.0f;
        }
    };

    uint32_t tile = r0;
    bool warm = needs_warm(r0, false);
    int par = 0;                                                    // carry buffer phase A of this pass writes
    prep_tile(tile, warm, true);
#pragma unroll
    for (int s2 = 0; s2 < kLoads; ++s2) issue_load(s2);
    while (tile < r1) {
        // (no barrier here: xs was last read in phase A, ys in phase B of the previous pass; a wave writes xs before, and
        // ys after, the next barrier, which every wave reaches after its own phase B)
        {
            if constexpr (PAIR) {
                float* dst = xs + 2 * (tid + opaque_zero());
                if (first) {
#pragma unroll
                    for (int cc = 0; cc < kCols; ++cc) {
                        float m0, m1;
                        staged_pair(cc, m0, m1);
                        dst[cc * g.seg_stride] = m0;
                        dst[cc * g.seg_stride + 1] = m1;
                    }
                }
            } else {
                float* dst = xs + tid + opaque_zero();
                if (first) {
#pragma unroll
                    for (int cc = 0; cc < kCols; ++cc) dst[cc * g.seg_stride] = pf[cc * kPfSlots];
                }
                if (second) {
#pragma unroll
                    for (int cc = 0; cc < kCols; ++cc) dst[cc * g.seg_stride + nthreads] = pf[cc * kPfSlots + 1];
                }
            }
        }
        __syncthreads();
        const TileCtx tc = tile_ctx(tile, warm);
        // the pass after this one
        const uint32_t tile_n = warm ? tile : tile + 1;
        const bool live_n = tile_n < r1;
        const bool warm_n = warm ? false : (live_n ? needs_warm(tile_n, true) : false);
        prep_tile(live_n ? tile_n : tile, warm_n, live_n);

        // ---- phase A: resampled samples [y0, y0 + 5 120) into the FIR's column segments ----
        const int64_t y0 = tc.t_abs * y_per_tile;
        // (the lane's offsets, from a lane index the compiler cannot see through: hoisted out of the tile loop they cost
        // twenty registers next to 64 + 64 of fragments and staging -- 23 spilled, every reload behind the x requests)
        const int lz = lane + opaque_zero();
        const int colz_l = lz & 15, g4z = lz >> 4;
        const int base_full_a = colz_l * g.seg_stride + 16 * g4z + qp + g.shift;
        const int base_tail_a = colz_l * g.seg_stride + TAILA * g4z + 64 * FULLA + qp + g.shift;
        const int base_full_b = colz_l * g.seg_stride_b + 16 * g4z + wave_off_b;
        const int base_tail_b = colz_l * g.seg_stride_b + TAILB * g4z + 64 * FULLB + wave_off_b;
        const int y_lane = 16 * NB * (kG * colz_l) + 16 * wave + 4 * g4z;  // tile-relative resampled index of acc0[0]; acc1: + 160
        const int z_lane = 16 * (colz_l * P2 + wave * 2) + 4 * g4z;       // tile-relative FIR output of acc0[0]
        {
            // the FIR's history in front of the tile: zeros at a clip's start, else the previous pass's carry
            if (!warm && tid < km1) ys[tid] = tc.t_abs > 0 ? cw[(par ^ 1) * kCarry + tid] : 0.0f;
            const int64_t y_left = tc.n_res - y0;
            const int y_hi = y_left > y_per_tile ? y_per_tile : y_left < 0 ? 0 : (int)y_left;
            const float* b0p = xs + base_full_a;
            const float* t0p = xs + base_tail_a;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            chain2<NKA>(a1, b0p, b0p + g.S, t0p, t0p + g.S, acc0, acc1, [&](int t) {
                constexpr int kSpan = (NKA + 1) / 2;
                if (t < kSpan) {
#pragma unroll
                    for (int s2 = (kLoads * t) / kSpan; s2 < (kLoads * (t + 1)) / kSpan; ++s2) issue_load(s2);
                }
            });
            // branch-free scatter: position s of column segment c holds the FIR input with tile-relative index
            // c colz - (K - 1) + s; a sample on a segment's first K - 1 positions is the previous segment's tail too; the
            // tile's last K - 1 samples ("column 16") are the carry.  What belongs nowhere goes to a spare word per lane.
            const int spare = kCols * g.seg_stride_b + lane;
            float* cwp = cw + par * kCarry;
            auto put_y = [&](int yi, float v) {
                v = yi < y_hi ? v : 0.0f;
                const int e = yi + km1;
                const int c = e / colz;
                const int pos = e - c * colz;
                const int at = c * g.seg_stride_b + pos;
                float* own = c < kCols ? ys + at : cwp + pos;           // c == 16 (pos < K - 1): the carry
                *own = v;
                ys[(c >= 1 && pos + colz < g.seg_len_b) ? at - g.seg_stride_b + colz : spare] = v;
            };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                put_y(y_lane + j, acc0[j]);
                put_y(y_lane + 16 * NB + j, acc1[j]);
            }
        }
        __syncthreads();

        // ---- phase B: the FIR over the segments, 20 block periods per column ----
        if (!warm) {
            float* zc = tc.zp + y0;
            const int64_t lo_left = tc.m_lo - y0, hi_left = tc.m_hi - y0;
            const int z_lo = lo_left <= 0 ? 0 : lo_left > y_per_tile ? y_per_tile : (int)lo_left;
            const int z_hi = hi_left <= 0 ? 0 : hi_left > y_per_tile ? y_per_tile : (int)hi_left;
            const bool whole = z_lo == 0 && z_hi == y_per_tile && ((reinterpret_cast<uintptr_t>(zc) >> 2) & 3) == 0;
            const float* b0p = ys + base_full_b;
            const float* t0p = ys + base_tail_b;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            chain2<NKB>(a2, b0p, b0p + 16, t0p, t0p + 16, acc0, acc1, [&](int) {});
            const int o0 = z_lane, o1 = o0 + 16;
            if (whole) {
                *reinterpret_cast<f32x4*>(zc + o0) = acc0;
                *reinterpret_cast<f32x4*>(zc + o1) = acc1;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (o0 + j >= z_lo && o0 + j < z_hi) zc[o0 + j] = acc0[j];
                    if (o1 + j >= z_lo && o1 + j < z_hi) zc[o1 + j] = acc1[j];
                }
            }
        }
        par ^= 1;
        tile = tile_n;
        warm = warm_n;
    }
}

// ---- the wire path: resampler -> PCM16 -> base64 in one launch (banded.hpp WirePlan) ----------------------------
// Phase A is resfir_kernel's (same staging of the four PCM formats, same chain pair per wave), with the columns 15
// block periods apart instead of 16: wave 7's second chain (period 15) overlaps the next column and is dropped.  The
// accumulators are quantised where they are (websocket.rs:248), packed little-endian (:249) and written into an LDS
// image of the tile's 3 840 samples; behind the second barrier every thread takes whole base64 groups (6 samples = 12
// bytes = three dwords of the image -> 16 characters, websocket.rs:254) and stores them where consecutive lanes
// write consecutive 16 bytes.  With messages (frame > 0) a group never crosses a message: the last group of a message
// is short and carries the '=' padding of that message.
struct WireGeom {
    const void* x;
    char* text;
    int16_t* pcm16;                  // optional
    const float* afrag_a;
    const WireTile* tiles;           // ragged batch, or null
    int64_t x_len, x_stride, text_stride, pcm_stride, n_res;
    int64_t tiles_per_clip, n_tiles;
    int lo_off, S, seg_len, seg_stride, shift;
    int Gv, frame, gpf, row_len;     // gpf = groups per message (ceil(frame / 6)); frame == 0: one string per clip
    int img_off;                     // floats
    int text_align;                  // 16, 8, 4 or 1
    float* energy;                   // optional (LEVEL instances): per message, vad.rs:157-168 / max |x|
    float* peak;
    int64_t msg_stride;
    int fimg_off;                    // floats: the tile's samples as f32, [16 columns][16 Gv + 4]
};

struct WireCtx {
    const char* xp;
    char* tp;                        // the clip's first character
    int16_t* pp;                     // the clip's first PCM16 sample (or null)
    int64_t n_res, t_abs, msg0;
    uint32_t x_bytes;
};

// LEVEL: also the messages' VAD energy and peak, from an f32 image of the tile beside the PCM16 one.  The reference's energy
// is a LEFT fold in f32 (`frame.iter().map(|&x| x * x).sum()`, vad.rs:163-165): one lane folds one message, in order, no FMA.
// In the fused path the 16 kHz f32 signal exists nowhere else, so this is the only place a VAD can look at it.
constexpr int kLevelWave = 7;         // its second chain (block period 15) is dropped anyway: the lightest wave folds
template <int NKA, int FMT, bool LEVEL = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void wire_kernel(const WireGeom g)
{
    using F = FusedFmt<FMT>;
    constexpr bool PAIR = F::kPair;
    constexpr int BPF = F::kBytesPerFrame;
    extern __shared__ __attribute__((aligned(16))) float xs[];
    // [16 columns][8 Gv + 2] dwords = 256 Gv little-endian i16: a column's 8 Gv dwords and two of padding, so that the 16
    // columns of a ds_write_b64 start 26 banks apart (16 different bank pairs; unpadded, 120 = 24 mod 32 dwords: four-way)
    unsigned* img = reinterpret_cast<unsigned*>(xs + g.img_off);
    const int img_col = 8 * g.Gv;                                   // dwords of samples per column
    constexpr int FULLA = NKA / 16, TAILA = NKA % 16;
    constexpr int nthreads = 512;
    constexpr int kGpw = 2;                                         // one chain pair per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_off_a = wave * kGpw * g.S;
    const int y_per_tile = 256 * g.Gv;
    const int col_span = g.Gv * g.S;
    const bool first = PAIR ? 2 * tid < g.seg_len + g.shift : tid < g.seg_len;
    const bool second = !PAIR && tid + nthreads < g.seg_len;
    const uint32_t tpc = (uint32_t)g.tiles_per_clip, n_tiles = (uint32_t)g.n_tiles;

    auto tile_ctx = [&](uint32_t tile) {
        WireCtx c;
        if (g.tiles) {
            typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
            typedef const __attribute__((address_space(4))) u32x8* cptr;
            const cptr rec = reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(g.tiles)) + 2 * (size_t)__builtin_amdgcn_readfirstlane(tile);
            const u32x8 r = rec[0];
            const int64_t x_off = (int64_t)(((unsigned long long)r[1] << 32) | r[0]);
            const int64_t t_off = (int64_t)(((unsigned long long)r[3] << 32) | r[2]);
            const int64_t p_off = (int64_t)(((unsigned long long)r[5] << 32) | r[4]);
            c.xp = static_cast<const char*>(g.x) + x_off * BPF;
            c.tp = g.text + t_off;
            c.pp = g.pcm16 ? g.pcm16 + p_off : nullptr;
            c.x_bytes = r[6] * (uint32_t)BPF;
            c.n_res = (int64_t)r[7];
            const u32x8 r2 = rec[1];
            c.t_abs = (int64_t)r2[0];
            c.msg0 = (int64_t)(((unsigned long long)r2[2] << 32) | r2[1]);
        } else {
            const uint32_t clip = tile / tpc;
            c.xp = static_cast<const char*>(g.x) + (int64_t)clip * g.x_stride * BPF;
            c.tp = g.text + (int64_t)clip * g.text_stride;
            c.pp = g.pcm16 ? g.pcm16 + (int64_t)clip * g.pcm_stride : nullptr;
            c.x_bytes = (uint32_t)(g.x_len * BPF);
            c.n_res = g.n_res;
            c.t_abs = tile - clip * tpc;
            c.msg0 = (int64_t)clip * g.msg_stride;
        }
        return c;
    };

    // the resampler's fragments stay in registers for the whole launch.  Waited for HERE, with a wait the compiler sees:
    // otherwise its wait-count pass carries these loads as possibly pending around the tile loop and waits for the
    // x requests of the tile in flight in front of every chain.
    float a1[NKA];
#pragma unroll
    for (int t = 0; t < NKA; ++t) a1[t] = g.afrag_a[t * 64 + lane];
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0)
#pragma unroll
    for (int t = 0; t < NKA; ++t) asm volatile("" : "+v"(a1[t]));

    constexpr uint32_t kOob = 0xFFF00000u;
    constexpr int kLoads = PAIR ? kCols : kCols * kPfSlots;          // load instructions per thread and tile
    constexpr int kRegs = kCols * F::kRegsPerCol;
    float pf[kRegs];
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[kPfSlots];
    uint32_t x_frames = 0;           // FMT 3: frames of the clip whose tile is staged next (see staged_pair)
    auto prep_tile = [&](uint32_t tile, bool live) {
        const WireCtx c = tile_ctx(tile);
        // x index of the first staged sample of the tile: q(y0) + lo_off - shift, y0 = t y_per_tile
        const int64_t base = c.t_abs * (int64_t)(16 * g.Gv) * g.S + g.lo_off - g.shift;
        // (mono i16, odd frame counts: see resfir_kernel)
        if constexpr (FMT == 3) x_frames = c.x_bytes >> 1;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(c.xp), 0, (int)(FMT == 3 ? (c.x_bytes + 2u) & ~3u : c.x_bytes),
                                                 0x00020000);
        if constexpr (PAIR) {
            const uint32_t v0 = ((uint32_t)base + 2u * (uint32_t)(tid + opaque_zero())) * (uint32_t)BPF;
            voff[0] = live && first ? v0 : kOob;
        } else {
            const uint32_t v0 = ((uint32_t)base + (uint32_t)(tid + opaque_zero())) * 4u;
            voff[0] = live && first ? v0 : kOob;
            voff[1] = live && second ? v0 + (uint32_t)nthreads * 4u : kOob;
        }
    };
    auto issue_load = [&](int s) {
        if constexpr (FMT == 0) {                                   // s = h * 16 + cc
            const int cc = s % kCols, h = s / kCols;
            pf[cc * kPfSlots + h] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(voff[h] + (uint32_t)(cc * col_span) * 4u), 0, AF_X_AUX));
        } else if constexpr (FMT == 1) {                            // s = cc
            const f32x4 v = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(voff[0] + (uint32_t)(s * col_span) * 8u), 0, AF_X_AUX));
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[s * 4 + i] = v[i];
        } else if constexpr (FMT == 2) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 v = __builtin_bit_cast(
                f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(voff[0] + (uint32_t)(s * col_span) * 4u), 0, AF_X_AUX));
            pf[s * 2] = v[0];
            pf[s * 2 + 1] = v[1];
        } else {
            pf[s] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(voff[0] + (uint32_t)(s * col_span) * 2u), 0, AF_X_AUX));
        }
    };
    auto staged_pair = [&](int cc, float& m0, float& m1) {
        auto lo16 = [](float w) { return (float)(int16_t)(__float_as_uint(w) & 0xffffu) * (1.0f / 32768.0f); };
        auto hi16 = [](float w) { return (float)(int16_t)(__float_as_uint(w) >> 16) * (1.0f / 32768.0f); };
        if constexpr (FMT == 1) {
            m0 = (pf[cc * 4] + pf[cc * 4 + 1]) * 0.5f;
            m1 = (pf[cc * 4 + 2] + pf[cc * 4 + 3]) * 0.5f;
        } else if constexpr (FMT == 2) {
            m0 = (lo16(pf[cc * 2]) + hi16(pf[cc * 2])) * 0.5f;
            m1 = (lo16(pf[cc * 2 + 1]) + hi16(pf[cc * 2 + 1])) * 0.5f;
        } else {
            m0 = lo16(pf[cc]);
            const uint32_t b1 = voff[0] + (uint32_t)(cc * col_span) * 2u + 2u;
            m1 = b1 < 2u * x_frames ? hi16(pf[cc]) : 0.0f;
        }
    };

    if constexpr (LEVEL) {
        if (tid < 64) reinterpret_cast<unsigned*>(xs + g.fimg_off + kCols * (16 * g.Gv + 4))[tid] = 0u;    // (visible behind the first barrier)
    }
    uint32_t tile = blockIdx.x;
    if (tile < n_tiles) {
        prep_tile(tile, true);
#pragma unroll
        for (int s2 = 0; s2 < kLoads; ++s2) issue_load(s2);
    }
    for (; tile < n_tiles; tile += gridDim.x) {
        // No barrier here: xs was last read by the previous tile's chains and the image by its encoding threads; a wave
        // writes xs before, and the image after, the next barrier, which every wave reaches after its own encoding.
        {
            if constexpr (PAIR) {
                float* dst = xs + 2 * (tid + opaque_zero());
                if (first) {
#pragma unroll
                    for (int cc = 0; cc < kCols; ++cc) {
                        float m0, m1;
                        staged_pair(cc, m0, m1);
                        dst[cc * g.seg_stride] = m0;
                        dst[cc * g.seg_stride + 1] = m1;
                    }
                }
            } else {
                float* dst = xs + tid + opaque_zero();
                if (first) {
#pragma unroll
                    for (int cc = 0; cc < kCols; ++cc) dst[cc * g.seg_stride] = pf[cc * kPfSlots];
                }
                if (second) {
#pragma unroll
                    for (int cc = 0; cc < kCols; ++cc) dst[cc * g.seg_stride + nthreads] = pf[cc * kPfSlots + 1];
                }
            }
        }
        __syncthreads();
        const WireCtx tc = tile_ctx(tile);
        {
            const bool live = tile + gridDim.x < n_tiles;
            prep_tile(live ? tile + gridDim.x : tile, live);
        }

        // ---- resampled samples [y0, y0 + 256 Gv) -> PCM16 image ----
        const int64_t y0 = tc.t_abs * y_per_tile;
        const int64_t y_left = tc.n_res - y0;
        const int y_hi = y_left > y_per_tile ? y_per_tile : y_left < 0 ? 0 : (int)y_left;
        // (the lane's offsets from a lane index the compiler cannot see through: hoisted out of the tile loop they sit next
        // to 28 fragment and 64 staging registers and spill -- every reload then waits behind the x requests in flight)
        const int lz = lane + opaque_zero();
        const int col = lz & 15, g4 = lz >> 4;
        const int base_full_a = col * g.seg_stride + 16 * g4 + wave_off_a + g.shift;
        const int base_tail_a = col * g.seg_stride + TAILA * g4 + 64 * FULLA + wave_off_a + g.shift;
        const int y_lane = 16 * (col * g.Gv + wave * kGpw) + 4 * g4;    // tile-relative resampled index of acc0[0]
        {
            const float* b0p = xs + base_full_a;
            const float* t0p = xs + base_tail_a;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            chain2<NKA>(a1, b0p, b0p + g.S, t0p, t0p + g.S, acc0, acc1, [&](int t) {
                // all of the next tile's requests in the first half of the chain
                constexpr int kSpan = (NKA + 1) / 2;
                if (t < kSpan) {
#pragma unroll
                    for (int s2 = (kLoads * t) / kSpan; s2 < (kLoads * (t + 1)) / kSpan; ++s2) issue_load(s2);
                }
            });
            // websocket.rs:248-249; samples behind the clip's end are zero bytes (the padding bits of its last group)
            auto quads = [&](const f32x4& acc, int yi, unsigned& lo, unsigned& hi) {
                unsigned q[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] = yi + j < y_hi ? (unsigned)pcmdev::quant1(acc[j]) & 0xffffu : 0u;
                lo = q[0] | (q[1] << 16);
                hi = q[2] | (q[3] << 16);
            };
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            unsigned lo, hi;
            unsigned* ic = img + col * (img_col + 2) + 8 * wave * kGpw + 2 * g4;
            quads(acc0, y_lane, lo, hi);
            *reinterpret_cast<u32x2*>(ic) = u32x2{lo, hi};
            if (wave * kGpw + 1 < g.Gv) {                            // wave-uniform: period 15 belongs to the next column
                quads(acc1, y_lane + 16, lo, hi);
                *reinterpret_cast<u32x2*>(ic + 8) = u32x2{lo, hi};
            }
            if constexpr (LEVEL) {
                // The samples' SQUARES as f32 (zero behind the clip's end) -- the products of vad.rs:163-165, one rounding
                // each, computed by every lane for its own samples; what is left to the fold is the sum.  Columns 16 Gv + 4
                // floats apart: the eight lanes of a ds_write_b128 group start 20 banks apart, no two on the same four.
                float* fc = xs + g.fimg_off + col * (16 * g.Gv + 4) + 16 * wave * kGpw + 4 * g4;
                unsigned* pkb = reinterpret_cast<unsigned*>(xs + g.fimg_off + kCols * (16 * g.Gv + 4));   // [64] max |x| bits per message
                auto squares = [&](const f32x4& acc, int yi, float& m) {
                    f32x4 p;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = yi + j < y_hi ? acc[j] : 0.0f;
                        p[j] = v * v;
                        m = fmaxf(m, fabsf(v));
                    }
                    return p;
                };
                // peak: the 16 outputs of a block (the four g4 lanes of a column) lie in one message (frame % 16 == 0):
                // max over those lanes, then one LDS atomic per block (|x| >= 0: its bits order like unsigned integers)
                auto block_peak = [&](float m, int yi, bool on) {
                    m = fmaxf(m, __shfl_xor(m, 16));
                    m = fmaxf(m, __shfl_xor(m, 32));
                    if (on && g4 == 0 && yi < y_hi) atomicMax(pkb + yi / g.frame, __float_as_uint(m));
                };
                float m0 = 0.0f, m1 = 0.0f;
                *reinterpret_cast<f32x4*>(fc) = squares(acc0, y_lane, m0);
                const bool second_on = wave * kGpw + 1 < g.Gv;
                const f32x4 p1 = squares(acc1, y_lane + 16, m1);
                if (second_on) *reinterpret_cast<f32x4*>(fc + 16) = p1;
                block_peak(m0, y_lane, true);
                block_peak(m1, y_lane + 16, second_on);
            }
        }
        __syncthreads();

        if constexpr (LEVEL) {
            // ---- the messages' energy: ONE wave, message f of the tile on lane f, a left fold over the message's squares
            // (vad.rs:163-165 `.sum()`: in order, f32).  The other waves go on to the base64 groups and the next tile's staging.
            if (wave == kLevelWave) {
                const int n_msg = y_per_tile / g.frame;
                const int f = lane;
                const int64_t fa = y0 / g.frame + f;
                unsigned* pkb = reinterpret_cast<unsigned*>(xs + g.fimg_off + kCols * (16 * g.Gv + 4));
                if (f < n_msg) {
                    // A message and a column (240 samples) both start on multiples of 80 samples (frame % 80 == 0): the fold
                    // walks chunks of 20 quads that never cross a column's padding, five 16-byte reads in flight at a time
                    // (one read per iteration left every add waiting for an LDS round trip: 8.2 ms for the launch).
                    const int colf = 16 * g.Gv;
                    const int s0 = f * g.frame;
                    float e = 0.0f;
                    for (int i = 0; i < g.frame; i += 80) {
                        const int s = s0 + i, c = s / colf;
                        const f32x4* q = reinterpret_cast<const f32x4*>(xs + g.fimg_off + c * (colf + 4) + (s - c * colf));
#pragma unroll
                        for (int h = 0; h < 4; ++h) {                 // (five reads = 20 registers: the next tile's 64 staging registers are live)
                            f32x4 p[5];
#pragma unroll
                            for (int k = 0; k < 5; ++k) p[k] = q[5 * h + k];
#pragma unroll
                            for (int k = 0; k < 5; ++k) { e = e + p[k][0]; e = e + p[k][1]; e = e + p[k][2]; e = e + p[k][3]; }
                        }
                    }
                    const unsigned mb = pkb[f];
                    pkb[f] = 0u;                                     // for the next tile (its atomics come behind the next barrier)
                    if ((fa + 1) * g.frame <= tc.n_res) {             // only whole messages exist (specs/0002-design.md:1543-1546)
                        if (g.energy) g.energy[tc.msg0 + fa] = e / (float)g.frame;      // vad.rs:167
                        if (g.peak) g.peak[tc.msg0 + fa] = __uint_as_float(mb);
                    }
                }
            }
        }

        // ---- base64: whole groups of 6 samples per thread ----
        {
            const int n_items = g.frame ? (y_per_tile / g.frame) * g.gpf : y_per_tile / 6;
            // (LEVEL: the folding wave, the last one, leaves the groups to the other seven)
            constexpr int kEncThreads = LEVEL ? nthreads - 64 : nthreads;
            for (int it = tid; it < n_items && tid < kEncThreads; it += kEncThreads) {
                int s0, cnt;                                         // first sample of the group in the tile, real samples
                int64_t toff;                                        // its first character in the clip's text
                bool send = true;                                    // the group's characters are part of the text
                if (g.frame) {
                    const int f = it / g.gpf, j = it - f * g.gpf;
                    s0 = f * g.frame + 6 * j;
                    const int left = g.frame - 6 * j, have = y_hi - s0;
                    cnt = left < 6 ? left : 6;
                    cnt = have < cnt ? (have < 0 ? 0 : have) : cnt;
                    // only whole messages are sent (specs/0002-design.md:1543-1546); the samples of a trailing partial
                    // batch still go to the PCM16 row
                    const int64_t fa = y0 / g.frame + f;
                    send = (fa + 1) * g.frame <= tc.n_res;
                    toff = fa * g.row_len + 16 * j;
                } else {
                    s0 = 6 * it;
                    const int left = y_hi - s0;
                    cnt = left < 0 ? 0 : left < 6 ? left : 6;
                    toff = (y0 / 6 + it) * 16;
                }
                if (cnt <= 0) continue;
                // s0 is even (frame is); a message's group may straddle two columns of the image
                const int d0 = s0 >> 1;
                auto at = [&](int d) { return img[d + 2 * (d / img_col)]; };
                unsigned w0 = at(d0), w1 = at(d0 + 1), w2 = at(d0 + 2);
                if (g.frame && cnt < 6) {                            // the samples behind a message's end belong to the next one
                    if (cnt <= 4) w2 = 0u;
                    if (cnt <= 2) w1 = 0u;
                }
                const uint4 o = pcmdev::b64_encode_words(w0, w1, w2);
                char* dst = tc.tp + toff;
                if (!send) {
                } else if (cnt == 6) {
                    if (g.text_align == 16) {
                        *reinterpret_cast<uint4*>(dst) = o;
                    } else if (g.text_align == 8) {
                        reinterpret_cast<uint2*>(dst)[0] = uint2{o.x, o.y};
                        reinterpret_cast<uint2*>(dst)[1] = uint2{o.z, o.w};
                    } else if (g.text_align == 4) {
                        unsigned* d4 = reinterpret_cast<unsigned*>(dst);
                        d4[0] = o.x; d4[1] = o.y; d4[2] = o.z; d4[3] = o.w;
                    } else {
                        const unsigned ow[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                        for (int c = 0; c < 16; ++c) dst[c] = (char)((ow[c >> 2] >> (8 * (c & 3))) & 0xffu);
                    }
                } else {
                    pcmdev::b64_store_tail(dst, o, 2u * (unsigned)cnt);
                }
                if (tc.pp) {                                         // the samples themselves, if wanted
                    int16_t* d16 = tc.pp + y0 + s0;
                    if (cnt == 6 && (reinterpret_cast<uintptr_t>(d16) & 3) == 0) {
                        unsigned* d = reinterpret_cast<unsigned*>(d16);
                        d[0] = w0; d[1] = w1; d[2] = w2;
                    } else {
                        const unsigned ww[3] = {w0, w1, w2};
                        for (int e = 0; e < cnt; ++e) d16[e] = (int16_t)(ww[e >> 1] >> (16 * (e & 1)));
                    }
                }
            }
        }
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
            if (g.buffer_stores) hipLaunchKernelGGL((banded_kernel<NK, 5, 1, false, true>), dim3(grid), dim3(block), smem, s, g);
            else hipLaunchKernelGGL((banded_kernel<NK, 5, 1>), dim3(grid), dim3(block), smem, s, g);
            return;
        }
    }
    if (g.accumulate) {
        hipLaunchKernelGGL((banded_kernel<NK, 4, 1, true>), dim3(grid), dim3(block), smem, s, g);
        return;
    }
    hipLaunchKernelGGL((banded_kernel<NK, 4, 1>), dim3(grid), dim3(block), smem, s, g);
}

}  // namespace

namespace {
constexpr int kStreamC = 2;                    // column sets per tile of the streamed kernel (instances below)
constexpr int kStreamNk[] = {20, 24, 28, 36};

// Ratios with more than 16 block matrices (or block periods beyond an LDS segment): the streamed kernel.
Plan make_stream_plan(Plan p, uint64_t nb, uint64_t S)
{
    // S: an idle lane's byte offset is kOob = 0xFFF00000 plus up to 2 (16 C - 1) S 4 bytes of segment offset and must
    // stay below 2^32 (out of range for the hardware check), so S is bounded well below 2^20 / (32 C) samples
    if (p.ch != 1 || nb > 4096 || S > 6000) return p;
    uint64_t d15 = 0;
    for (uint64_t b = 0; b < nb; ++b) {
        const uint64_t d = ((16 * b * p.M) % p.L + 15ull * p.M) / p.L;
        if (d > d15) d15 = d;
    }
    const uint64_t W = (uint64_t)p.T + d15;
    const uint64_t need = 16 * (W / 64) + ((W % 64) + 3) / 4;
    int nk = 0;
    for (int cand : kStreamNk)
        if ((uint64_t)cand >= need) { nk = cand; break; }
    if (!nk) return p;
    // the largest range of matrices (a multiple of the 8 waves) whose windows fit one 512-sample segment
    auto q16 = [&](uint64_t pp) { return (16 * pp * p.M) / p.L; };
    int best = 0, best_len = 0;
    for (int P = AF_STREAM_PMAX; P >= 8; P -= 8) {
        uint64_t span = 0;
        for (uint64_t p0 = 0; p0 < nb; p0 += P) {
            const uint64_t last = std::min<uint64_t>(p0 + P, nb) - 1;
            span = std::max(span, q16(last) - q16(p0));
        }
        if (span + 4 * nk <= 511) { best = P; best_len = (int)span + 4 * nk; break; }
    }
    if (!best) return p;
    p.nb = (int)nb;
    p.S = (int)S;
    p.NK = nk;
    p.streamed = true;
    p.P = best;
    p.C = kStreamC;
    p.nwaves = 8;
    p.wpp = 1;
    p.G = 0;
    p.seg_len = best_len;
    p.seg_stride = (best_len + 2) | 1;         // odd (bank-conflict-free operand reads); a lead-in word and a pair's odd end
    p.out_per_tile = (int64_t)256 * best * kStreamC;
    p.smem = (size_t)16 * kStreamC * p.seg_stride * sizeof(float);
    // deferred stores (banded_stream_kernel, OB): an LDS image of the tile's outputs behind the segments, rows padded by
    // four floats (the chains' 16-byte writes of one column group then hit 16 different bank quads) -- where two
    // workgroups per CU still fit
    if (AF_STREAM_OB && best <= 16) {
        const int ob_off = (16 * kStreamC * p.seg_stride + 3) & ~3;
        const int ob_stride = best * 16 + 4;
        const size_t total = ((size_t)ob_off + (size_t)16 * kStreamC * ob_stride) * sizeof(float);
        if (total <= 80 * 1024) {
            p.ob_off = ob_off;
            p.ob_stride = ob_stride;
            p.smem = total;
        }
    }
    p.ok = true;
    return p;
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
    if (nb > 16 || S > (uint64_t)kMaxSeg) return make_stream_plan(p, nb, S);
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
    // stereo staging holds (L, R) pairs: 128 registers (four waves per SIMD), i.e. ONE workgroup per CU for the shapes
    // with more than eight waves (44.1 k -> 16 k: ten) -- still one pass over the stereo PCM instead of to_mono + a
    // mono resampler pass
    if (ch == 2 && (p.nwaves > 16 || p.NK > 32)) return p;
    // mono x is staged in sample pairs (lead-in word, odd end) except in the shapes launch() runs at five waves per SIMD
    const bool pairs = ch == 1 && AF_BANDED_PAIRS && !(p.nwaves > 8 && p.NK <= 32);
    p.pairs = pairs;
    const int seg_cap = (kPfSlots * nthreads < kMaxSeg ? kPfSlots * nthreads : kMaxSeg) - (pairs ? 1 : 0);
    const int qp_last = (int)((16ull * (nb - 1) * M) / L);
    const int unit = 2 * p.wpp;                              // every wave runs chains in pairs
    int G = 0;
    for (int cand = unit; (int64_t)(cand - 1) * p.S + qp_last + 4 * p.NK <= seg_cap; cand += unit) G = cand;
    if (!G) return p;
    p.G = G;
    p.seg_len = (G - 1) * p.S + qp_last + 4 * p.NK;
    p.seg_stride = pairs ? (p.seg_len + 2) | 1 : p.seg_len | 1;
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
                       afrag_dev, p.L, p.M, p.T, p.nb, p.NK, p.streamed ? 1 : 0);
    AF_HIP(hipGetLastError());
    if (p.streamed) {
        hipLaunchKernelGGL(build_qtab_kernel, dim3((p.nb + 256) / 256), dim3(256), 0, stream,
                           reinterpret_cast<int*>(afrag_dev + (size_t)p.nb * p.NK * 64), p.L, p.M, p.nb);
        AF_HIP(hipGetLastError());
    }
    return AF_OK;
}

namespace {
template <int NK>
af_status launch_stream_nk(const StreamGeom& g, unsigned grid, size_t smem, hipStream_t s)
{
    if (g.ob_stride) {
        static LdsAttrFlags attr_set;
        AF_HIP(allow_big_lds(banded_stream_kernel<NK, kStreamC, true>, 160 * 1024, attr_set));
        hipLaunchKernelGGL((banded_stream_kernel<NK, kStreamC, true>), dim3(grid), dim3(512), smem, s, g);
    } else {
        static LdsAttrFlags attr_set;
        AF_HIP(allow_big_lds(banded_stream_kernel<NK, kStreamC, false>, 160 * 1024, attr_set));
        hipLaunchKernelGGL((banded_stream_kernel<NK, kStreamC, false>), dim3(grid), dim3(512), smem, s, g);
    }
    AF_HIP(hipGetLastError());
    return AF_OK;
}

af_status launch_stream(const Plan& p, const float* afrag_dev, const float* x_dev, int64_t x_abs0, int64_t x_len,
                        int64_t x_stride, float* y_dev, int64_t y_stride, int64_t m0, int64_t n_out, int64_t n_clips,
                        hipStream_t stream)
{
    StreamGeom g{};
    g.x = x_dev;
    g.y = y_dev;
    g.afrag = afrag_dev;
    g.qtab = reinterpret_cast<const int*>(afrag_dev + (size_t)p.nb * p.NK * 64);
    g.x_abs0 = x_abs0;
    g.x_len = x_len;
    g.x_stride = x_stride;
    g.y_stride = y_stride;
    g.m0 = m0;
    g.n_out = n_out;
    const int64_t per_group = (int64_t)16 * p.C;                    // periods per tile
    const int64_t period_lo = (m0 / 16) / p.nb, period_hi = ((m0 + n_out - 1) / 16) / p.nb;
    g.pg0 = period_lo / per_group;
    g.n_pg = period_hi / per_group - g.pg0 + 1;
    g.n_pr = (p.nb + p.P - 1) / p.P;
    g.n_tiles = g.n_pg * g.n_pr * n_clips;
    if (g.n_tiles >= (1ll << 31)) return fail(AF_ERR_UNSUPPORTED, "banded::launch: too many tiles");
    // deferred stores address y with 32-bit byte offsets from output m0 of a clip; a tile reaches at most 16 * 16 C * nb
    // outputs past either end of [m0, m0 + n_out), and neither side may wrap into the range the hardware accepts
    // (and its 16-byte stores need 16-byte-aligned addresses: y itself, every clip's row and output m0 on that grid)
    const bool y_aligned = (reinterpret_cast<uintptr_t>(y_dev) & 15) == 0 && (y_stride & 3) == 0 && (m0 & 3) == 0;
    const bool ob = p.ob_stride != 0 && y_aligned && n_out + (int64_t)2 * 256 * p.C * p.nb < (1ll << 29);
    g.ob_off = ob ? p.ob_off : 0;
    g.ob_stride = ob ? p.ob_stride : 0;
    g.lo_off = p.c - (p.T - 1);
    g.nb = p.nb;
    g.S = p.S;
    g.P = p.P;
    g.C = p.C;
    g.seg_len = p.seg_len;
    g.seg_stride = p.seg_stride;
    int64_t grid = g.n_tiles;
    const int64_t cap = 2 * (int64_t)device_cu_count(current_device());
    if (grid > cap) grid = cap;
    switch (p.NK) {
    case 20: return launch_stream_nk<20>(g, (unsigned)grid, p.smem, stream);
    case 24: return launch_stream_nk<24>(g, (unsigned)grid, p.smem, stream);
    case 28: return launch_stream_nk<28>(g, (unsigned)grid, p.smem, stream);
    case 36: return launch_stream_nk<36>(g, (unsigned)grid, p.smem, stream);
    default: return fail(AF_ERR_UNSUPPORTED, "no streamed banded kernel for NK=%d", p.NK);
    }
}
}  // namespace

af_status launch(const Plan& p, const float* afrag_dev, const float* x_dev, int64_t x_abs0,
                 int64_t x_len, int64_t x_stride, float* y_dev, int64_t y_stride, int64_t m0,
                 int64_t n_out, int64_t n_clips, hipStream_t stream, bool accumulate)
{
    if (n_out <= 0 || n_clips <= 0) return AF_OK;
    if (!p.ok) return fail(AF_ERR_UNSUPPORTED, "banded::launch without a plan");
    if (x_len * p.ch >= (int64_t)kMaxClip)
        return fail(AF_ERR_UNSUPPORTED, "clips of %lld samples are beyond the banded kernel's 32-bit offsets",
                    (long long)x_len);
    if (p.streamed) {
        if (accumulate) return fail(AF_ERR_UNSUPPORTED, "banded::launch: the streamed kernel does not accumulate");
        return launch_stream(p, afrag_dev, x_dev, x_abs0, x_len, x_stride, y_dev, y_stride, m0, n_out, n_clips, stream);
    }
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
    // (16-byte stores at 16-byte-aligned addresses only: y itself, every clip's row and output m0 on that grid -- a
    // stream's chunks start anywhere and keep the predicated stores)
    const bool y_aligned = (reinterpret_cast<uintptr_t>(y_dev) & 15) == 0 && (y_stride & 3) == 0 && (m0 & 3) == 0;
    g.buffer_stores = AF_BANDED_BS && !accumulate && y_aligned && n_out + 2 * p.out_per_tile < (1ll << 29) ? 1 : 0;
    const int wg_per_cu = (2 * 64 * p.nwaves <= 2048 && 2 * p.smem <= 150 * 1024 && !(p.ch == 2 && p.nwaves > 8)) ? 2 : 1;
    int64_t grid = g.n_tiles;
    const int64_t n_cus = device_cu_count(current_device());
    if (grid > n_cus * wg_per_cu) grid = n_cus * wg_per_cu;
    const unsigned block = 64u * p.nwaves;
    const bool five = wg_per_cu == 2 && 2 * p.nwaves > 16;
    if (five && p.NK <= 32 && p.pairs)
        return fail(AF_ERR_UNSUPPORTED, "banded::launch: plan staged in pairs for a five-wave shape");
    if (!(five && p.NK <= 32) && p.ch == 1 && AF_BANDED_PAIRS && !p.pairs) {
        // a ten-wave plan (no pairs) that does not get two workgroups per CU after all: the four-wave instance would
        // stage pairs into segments laid out without the lead-in word
        return fail(AF_ERR_UNSUPPORTED, "banded::launch: %d-wave plan without room for two workgroups per CU", p.nwaves);
    }
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

// ---- fused resampler + FIR ------------------------------------------------------------------------
namespace {

template <int NKA, int NKB, int FMT>
af_status launch_resfir(const FusedGeom& g, unsigned grid, size_t smem, hipStream_t s)
{
    static LdsAttrFlags attr_set;
    AF_HIP(allow_big_lds(resfir_kernel<NKA, NKB, FMT>, 160 * 1024, attr_set));
    hipLaunchKernelGGL((resfir_kernel<NKA, NKB, FMT>), dim3(grid), dim3(512), smem, s, g);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

template <int NKA, int FMT>
af_status launch_resfir_b(int nkb, const FusedGeom& g, unsigned grid, size_t smem, hipStream_t s)
{
    switch (nkb) {
    case 20: return launch_resfir<NKA, 20, FMT>(g, grid, smem, s);
    case 36: return launch_resfir<NKA, 36, FMT>(g, grid, smem, s);
    default: return fail(AF_ERR_UNSUPPORTED, "no fused kernel for a FIR of %d steps", nkb);
    }
}

template <int NKB, int FMT>
af_status launch_resfir10(const FusedGeom& g, unsigned grid, size_t smem, hipStream_t s)
{
    static LdsAttrFlags attr_set;
    AF_HIP(allow_big_lds(resfir10_kernel<28, NKB, FMT>, 160 * 1024, attr_set));
    hipLaunchKernelGGL((resfir10_kernel<28, NKB, FMT>), dim3(grid), dim3(640), smem, s, g);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

af_status launch_resfir10_f(int fmt, int nkb, const FusedGeom& g, unsigned grid, size_t smem, hipStream_t s)
{
    if (nkb != 20 && nkb != 36) return fail(AF_ERR_UNSUPPORTED, "no fused kernel for a FIR of %d steps", nkb);
    switch (fmt) {
    case kFmtMonoF32: return nkb == 20 ? launch_resfir10<20, 0>(g, grid, smem, s) : launch_resfir10<36, 0>(g, grid, smem, s);
    case kFmtStereoF32: return nkb == 20 ? launch_resfir10<20, 1>(g, grid, smem, s) : launch_resfir10<36, 1>(g, grid, smem, s);
    case kFmtStereoI16: return nkb == 20 ? launch_resfir10<20, 2>(g, grid, smem, s) : launch_resfir10<36, 2>(g, grid, smem, s);
    case kFmtMonoI16: return nkb == 20 ? launch_resfir10<20, 3>(g, grid, smem, s) : launch_resfir10<36, 3>(g, grid, smem, s);
    default: return fail(AF_ERR_UNSUPPORTED, "fused kernel: unknown input format %d", fmt);
    }
}

template <int NKA>
af_status launch_resfir_f(int fmt, int nkb, const FusedGeom& g, unsigned grid, size_t smem, hipStream_t s)
{
    switch (fmt) {
    case kFmtMonoF32: return launch_resfir_b<NKA, 0>(nkb, g, grid, smem, s);
    case kFmtStereoF32: return launch_resfir_b<NKA, 1>(nkb, g, grid, smem, s);
    case kFmtStereoI16: return launch_resfir_b<NKA, 2>(nkb, g, grid, smem, s);
    case kFmtMonoI16: return launch_resfir_b<NKA, 3>(nkb, g, grid, smem, s);
    default: return fail(AF_ERR_UNSUPPORTED, "fused kernel: unknown input format %d", fmt);
    }
}

}  // namespace

FusedPlan make_fused_plan(const Plan& a, int K, int fmt)
{
    FusedPlan f;
    f.a = a;
    f.K = K;
    f.fmt = fmt;
    if (!a.ok || K < 1 || fmt < 0 || fmt > 3) return f;
    if (a.nb == 10 && a.wpp == 1 && a.G == 2 && a.NK == 28 && K - 1 <= kCarry) {
        // ten block matrices, two block periods per column (44.1 k -> 16 k): resfir10_kernel
        const int need = 16 * ((K + 15) / 64) + (((K + 15) % 64) + 3) / 4;
        f.NKB = need <= 20 ? 20 : need <= 36 ? 36 : 0;
        if (!f.NKB) return f;
        f.nbk = 10;
        f.lead_tiles = 1;
        f.H2 = 0;
        f.pad = 0;
        f.P2 = 20;                                             // ten waves x one chain pair
        f.seg_len_b = (f.P2 - 1) * 16 + 4 * f.NKB;
        f.seg_stride_b = f.seg_len_b | 1;
        f.z_per_tile = (int64_t)256 * f.P2;                    // = 16 G nb 16: 32 rounds of the ten matrices
        if (fmt != kFmtMonoF32) {
            if ((a.G * a.S) % 2) return f;                     // frame pairs need an even tile origin: 32 S per tile, 2 S per column
            f.shift = (a.c - (a.T - 1)) & 1;
            if (a.seg_len + f.shift > 2 * 640) return f;       // a thread stages one pair per column
            f.seg_stride_a = ((a.seg_len + f.shift + 1) / 2 * 2) | 1;
        } else {
            if (a.seg_len > kPfSlots * 640) return f;
            f.seg_stride_a = a.seg_len | 1;
        }
        f.carry_off = kCols * (f.seg_stride_a + f.seg_stride_b) + 64;          // behind the spare words
        f.smem = ((size_t)f.carry_off + 2 * kCarry) * sizeof(float);
        if (f.smem > 160 * 1024) return f;                     // one workgroup per CU
        f.ok = true;
        return f;
    }
    if (a.nb != 1 || a.nwaves != 8) return f;
    if (a.NK != 24 && a.NK != 28) return f;                    // the instances built below
    const int need = 16 * ((K + 15) / 64) + (((K + 15) % 64) + 3) / 4;
    f.NKB = need <= 20 ? 20 : need <= 36 ? 36 : 0;
    if (!f.NKB) return f;
    f.H2 = 16 * ((K - 1 + 15) / 16);
    f.pad = f.H2 - (K - 1);
    f.P2 = a.G - 1;
    if (a.G != 16 || f.H2 > 256) return f;                    // one chain pair per wave and phase
    f.seg_len_b = (a.G - 1) * 16 + 4 * f.NKB;
    f.seg_stride_b = f.seg_len_b | 1;
    if (f.seg_len_b - 16 * f.P2 > 16 * f.P2) return f;         // a sample lands in at most two segments
    // every window sample phase B reads is a resampled sample of this tile
    if (15 * 16 * f.P2 + f.pad + f.seg_len_b > 256 * a.G) return f;
    f.z_per_tile = (int64_t)256 * f.P2;
    if (fmt != kFmtMonoF32) {
        if (a.S % 2) return f;                                 // frame pairs need an even tile origin
        f.shift = (a.c - (a.T - 1)) & 1;
        if (a.seg_len + f.shift > 2 * 512) return f;           // a thread stages one pair per column
        f.seg_stride_a = ((a.seg_len + f.shift + 1) / 2 * 2) | 1;
    } else {
        f.seg_stride_a = a.seg_stride;
    }
    f.smem = ((size_t)kCols * (f.seg_stride_a + f.seg_stride_b) + 64) * sizeof(float);   // + one spare word per lane
    if (2 * f.smem > 160 * 1024) return f;                     // two workgroups per CU or not at all
    f.ok = true;
    return f;
}

bool fits(const FusedPlan& f, int64_t x_len)
{
    return f.ok && x_len * fused_bytes_per_frame(f.fmt) < (int64_t)kMaxClip * 4 && x_len < (1ll << 31);
}

int64_t fused_tile_base(const FusedPlan& f, int64_t t)
{
    const int64_t lo_off = f.a.c - (f.a.T - 1);
    if (f.nbk == 10) return t * (int64_t)(16 * f.a.G) * f.a.S + lo_off - f.shift;      // 16 columns x G periods of S inputs
    return (t * 16 * f.P2 - f.H2 / 16) * f.a.S + lo_off - f.shift;
}

af_status build_fused_fir_afrag(const FusedPlan& f, const float* taps_dev, float* afrag_dev, hipStream_t stream)
{
    const int n = f.NKB * 64;
    hipLaunchKernelGGL(build_afrag_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, taps_dev, afrag_dev, 1u, 1u,
                       f.K, 1, f.NKB, 0);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

af_status launch_fused(const FusedPlan& f, const float* afrag_a, const float* afrag_b, const FusedArgs& q,
                       hipStream_t stream)
{
    if (!f.ok) return fail(AF_ERR_UNSUPPORTED, "banded::launch_fused without a plan");
    FusedGeom g{};
    g.x = q.x;
    g.z = q.z;
    g.afrag_a = afrag_a;
    g.afrag_b = afrag_b;
    if (q.tiles) {
        if (q.n_tiles <= 0) return AF_OK;
        g.tiles = q.tiles;
        g.n_tiles = q.n_tiles;
        g.tiles_per_clip = 1;
    } else {
        if (q.n_out <= 0 || q.n_clips <= 0) return AF_OK;
        if (!fits(f, q.x_len))
            return fail(AF_ERR_UNSUPPORTED, "clips of %lld frames are beyond the fused kernel's 32-bit offsets",
                        (long long)q.x_len);
        if (f.fmt != kFmtMonoF32 && (q.x_abs0 & 1))
            return fail(AF_ERR_INVALID_ARG, "banded::launch_fused: x must start on an even frame in the pair formats");
        g.x_abs0 = q.x_abs0;
        g.x_len = q.x_len;
        g.x_stride = q.x_stride;
        g.z_stride = q.z_stride;
        g.n_res = q.n_res;
        g.m0 = q.m0;
        g.n_out = q.n_out;
        g.tile0 = q.m0 / f.z_per_tile;
        g.tiles_per_clip = (q.m0 + q.n_out - 1) / f.z_per_tile - g.tile0 + 1;
        g.n_tiles = g.tiles_per_clip * q.n_clips;
    }
    if (g.n_tiles >= (1ll << 31)) return fail(AF_ERR_UNSUPPORTED, "banded::launch_fused: too many tiles");
    g.lo_off = f.a.c - (f.a.T - 1);
    g.S = f.a.S;
    g.G = f.a.G;
    g.gpw = f.a.G / f.a.wpp;
    g.seg_len = f.a.seg_len;
    g.seg_stride = f.seg_stride_a;
    g.shift = f.shift;
    g.H2 = f.H2;
    g.pad = f.pad;
    g.P2 = f.P2;
    g.seg_len_b = f.seg_len_b;
    g.seg_stride_b = f.seg_stride_b;
    g.ys_off = kCols * f.seg_stride_a;
    g.inv_colz = 1.0f / (float)(16 * f.P2);
    if (f.nbk == 10) {
        g.L = f.a.L;
        g.M = f.a.M;
        g.K = f.K;
        g.carry_off = f.carry_off;
        int64_t grid10 = g.n_tiles;
        const int64_t cus = (int64_t)device_cu_count(current_device());          // one ten-wave workgroup per CU
        if (grid10 > cus) grid10 = cus;
        return launch_resfir10_f(f.fmt, f.NKB, g, (unsigned)grid10, f.smem, stream);
    }
    int64_t grid = g.n_tiles;
    // (diagnostics: af_debug_set("features.wg_per_cu", 1) also holds this launch to one workgroup per CU)
    const int64_t cap = (debug_knob(kKnobFeaturesWgPerCu) == 1 || q.wg_per_cu == 1 ? 1 : 2) *
                        (int64_t)device_cu_count(current_device());
    if (grid > cap) grid = cap;
    if (f.a.NK == 24) return launch_resfir_f<24>(f.fmt, f.NKB, g, (unsigned)grid, f.smem, stream);
    return launch_resfir_f<28>(f.fmt, f.NKB, g, (unsigned)grid, f.smem, stream);
}


// ---- the wire launch ---------------------------------------------------------------------------------------------
namespace {
template <int NKA>
af_status launch_wire_f(int fmt, bool level, const WireGeom& g, unsigned grid, size_t smem, hipStream_t s)
{
    auto go = [&](auto kern) -> af_status {
        static LdsAttrFlags attr_set;
        AF_HIP(allow_big_lds(kern, 160 * 1024, attr_set));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, s, g);
        AF_HIP(hipGetLastError());
        return AF_OK;
    };
    switch (fmt) {
    case kFmtMonoF32: return level ? go(wire_kernel<NKA, 0, true>) : go(wire_kernel<NKA, 0>);
    case kFmtStereoF32: return level ? go(wire_kernel<NKA, 1, true>) : go(wire_kernel<NKA, 1>);
    case kFmtStereoI16: return level ? go(wire_kernel<NKA, 2, true>) : go(wire_kernel<NKA, 2>);
    case kFmtMonoI16: return level ? go(wire_kernel<NKA, 3, true>) : go(wire_kernel<NKA, 3>);
    default: return fail(AF_ERR_UNSUPPORTED, "wire kernel: unknown input format %d", fmt);
    }
}
}  // namespace

WirePlan make_wire_plan(const Plan& a, int fmt, int frame_samples)
{
    WirePlan w;
    w.a = a;
    w.fmt = fmt;
    if (!a.ok || a.nb != 1 || a.nwaves != 8 || a.G != 16 || fmt < 0 || fmt > 3 || frame_samples < 0) return w;
    if (a.NK != 24 && a.NK != 28) return w;                    // the instances built above
    w.Gv = a.G - 1;
    w.y_per_tile = (int64_t)256 * w.Gv;
    // a message never straddles two tiles, a group never straddles two messages' dwords
    if (frame_samples && (w.y_per_tile % frame_samples != 0 || frame_samples % 2 != 0)) return w;
    w.frame = frame_samples;
    w.row_len = 4 * ((2 * frame_samples + 2) / 3);
    if (fmt != kFmtMonoF32) {
        if (a.S % 2) return w;                                 // frame pairs need an even tile origin
        w.shift = (a.c - (a.T - 1)) & 1;
        if (a.seg_len + w.shift > 2 * 512) return w;           // a thread stages one pair per column
        w.seg_stride_a = ((a.seg_len + w.shift + 1) / 2 * 2) | 1;
    } else {
        w.seg_stride_a = a.seg_stride;
    }
    w.img_off = (kCols * w.seg_stride_a + 3) & ~3;             // 16-byte aligned
    w.smem = ((size_t)w.img_off + (size_t)kCols * (8 * w.Gv + 2)) * sizeof(float);    // image: padded columns
    w.fimg_off = (w.img_off + kCols * (8 * w.Gv + 2) + 3) & ~3;
    w.smem_level = ((size_t)w.fimg_off + (size_t)kCols * (16 * w.Gv + 4) + 64) * sizeof(float);   // + the messages' peak words
    if (2 * w.smem > 160 * 1024) return w;                     // two workgroups per CU or not at all
    w.ok = true;
    return w;
}

bool fits(const WirePlan& w, int64_t x_len)
{
    return w.ok && x_len * fused_bytes_per_frame(w.fmt) < (int64_t)kMaxClip * 4 && x_len < (1ll << 31);
}

af_status launch_wire(const WirePlan& w, const float* afrag_a, const WireArgs& q, hipStream_t stream)
{
    if (!w.ok) return fail(AF_ERR_UNSUPPORTED, "banded::launch_wire without a plan");
    WireGeom g{};
    g.x = q.x;
    g.text = q.text;
    g.pcm16 = q.pcm16;
    g.afrag_a = afrag_a;
    if (q.tiles) {
        if (q.n_tiles <= 0) return AF_OK;
        g.tiles = q.tiles;
        g.n_tiles = q.n_tiles;
        g.tiles_per_clip = 1;
    } else {
        if (q.n_res <= 0 || q.n_clips <= 0) return AF_OK;
        if (!fits(w, q.x_len))
            return fail(AF_ERR_UNSUPPORTED, "clips of %lld frames are beyond the wire kernel's 32-bit offsets", (long long)q.x_len);
        g.x_len = q.x_len;
        g.x_stride = q.x_stride;
        g.text_stride = q.text_stride;
        g.pcm_stride = q.pcm_stride;
        g.n_res = q.n_res;
        g.tiles_per_clip = (q.n_res + w.y_per_tile - 1) / w.y_per_tile;
        g.n_tiles = g.tiles_per_clip * q.n_clips;
    }
    if (g.n_tiles >= (1ll << 31)) return fail(AF_ERR_UNSUPPORTED, "banded::launch_wire: too many tiles");
    g.lo_off = w.a.c - (w.a.T - 1);
    g.S = w.a.S;
    g.seg_len = w.a.seg_len;
    g.seg_stride = w.seg_stride_a;
    g.shift = w.shift;
    g.Gv = w.Gv;
    g.frame = w.frame;
    g.gpf = (w.frame + 5) / 6;
    g.row_len = w.row_len;
    g.img_off = w.img_off;
    g.text_align = q.text_align;
    const bool level = q.energy || q.peak;
    if (level) {
        if (!w.frame || w.frame % 80 != 0 || w.y_per_tile / w.frame > 64)
            return fail(AF_ERR_UNSUPPORTED, "banded::launch_wire: energy / peak rows need messages of a multiple of 80 samples, at most 64 per tile of %lld (frame_samples %d)",
                        (long long)w.y_per_tile, w.frame);
        if (2 * w.smem_level > 160 * 1024) return fail(AF_ERR_UNSUPPORTED, "banded::launch_wire: no room for the f32 image beside two workgroups per CU");
        g.energy = q.energy;
        g.peak = q.peak;
        g.msg_stride = q.msg_stride;
        g.fimg_off = w.fimg_off;
    }
    int64_t grid = g.n_tiles;
    const int64_t cap = 2 * (int64_t)device_cu_count(current_device());
    if (grid > cap) grid = cap;
    const size_t smem = level ? w.smem_level : w.smem;
    if (w.a.NK == 24) return launch_wire_f<24>(w.fmt, level, g, (unsigned)grid, smem, stream);
    return launch_wire_f<28>(w.fmt, level, g, (unsigned)grid, smem, stream);
}

}  // namespace banded
}  // namespace af
