// features_v3.hip -- the fused  interleaved PCM -> to_mono -> Hann STFT(1024) -> |X|^2 -> mel  kernel for hop 256
// with the DFT on the matrix cores.
//
// Same contract as features_v2.hip / features_kernel.hip (to_mono: reference capture.rs:30-42; STFT / mel: docs/OPS.md,
// build-defined).  What differs is the transform: rounds 1-3 ran register FFTs on the vector pipe (features_v2.hip: 700
// VALU instructions per frame pair, no resource above 51 %, latency-bound); here the 1024-point DFT of the 16 frames of a
// unit is two matrix products on v_mfma_f32_16x16x32_f16, the 16 frames being the 16 columns:
//
//   n = 32 n1 + n2, k = k1 + 32 k2
//   stage 1 (per n2):  Y'[(k1, re/im)][f] = sum_n1 A1_n2[(k1, re/im)][n1] * x[256 f + 32 n1 + n2]
//            A1_n2 carries the Hann window w[n] AND the twiddle: w[n] * W1024^(n k1).  The input is real, so k1 = 0..16
//            suffice: Re k1 = 0..15 (row block 0), Im k1 = 1..15 plus the real, UNtwiddled row k1 = 16 (row block 1):
//            exactly 32 rows.
//   stage 2 (per k1):  X[k1 + 32 k2] = sum_n2 Y'[k1][n2] * W32^(n2 k2): the plain 32-point DFT matrix for every k1
//            (A2: rows (k2, cos | -sin), K = n2); real and imaginary parts of Y' are two products with the SAME
//            fragments (acc0 = A2 Yr, acc1 = A2 Yi;  Re X = acc0[cos row] - acc1[-sin row],  Im X = acc1[cos row] +
//            acc0[-sin row]).  k2 >= 16 gives bin 1024 - k, whose power is bin k's.  k1 = 0 (real input) and k1 = 16
//            (real input, matrix A2s with W64^n2 folded in) are one lighter work item.
//   There is no twiddle multiply, no window multiply, no spectrum split and no butterfly in the kernel.
//
// Precision: every operand is an f16 pair x = hi + 2^-11 lo' (hi = f16(x), lo' = f16((x - hi) 2^11): 22 significant
// bits), a product is three MFMAs (hi hi -> P1; hi lo', lo' hi -> P2; result P1 + 2^-11 P2, f32 accumulation).  The
// signal tile is scaled per unit by a power of two (max |x| -> [2^13, 2^14)) so that f16's exponent range sits where the
// unit's samples are; the scale leaves through the mel rows.  Measured against the float64 oracle: 2e-7 per frame
// (scripts/proto/dft_mfma_model.py is the numpy model of exactly this arithmetic).
//
// Structure: 512 threads = 8 waves, ONE workgroup per CU (256 registers per lane: the stage-1 fragments of a wave's four
// n2, 64 VGPRs, the stage-2 fragments, 32, and its mel fragments stay resident; 136 KB of LDS), a persistent grid over
// contiguous unit ranges, two barriers per unit:
//   first half    mel of the PREVIOUS unit (the banded f32 contraction of features_v2.hip on v_mfma_f32_16x16x4_f32, from
//                 bin-major power rows, branch-free), then
//                 stage 1 of this unit: wave w takes n2 = 4w .. 4w+3; B operand = 16 bytes of the f16 signal image
//                 T[n2][blk] (hi and lo' planes; frames overlap, so lanes (f, g) with equal f + g read the same address);
//                 results -> f16 pairs -> Y in LDS, [k1][f][Yr hi | Yi hi | Yr lo' | Yi lo'] (one ds_write_b64 = four n2), then
//                 the tile update for the NEXT unit into the second signal image (its piece of PCM arrived a pass ago and
//                 was turned into mono samples + maxima in the previous second half)
//   barrier 1
//   second half   stage 2: wave w takes k1 = 2w+1, 2w+2 (wave 7: k1 = 15 and the special item) -> power rows PT[bin][f];
//                 the prefetched piece of the tile after next -> mono samples and their maxima; the piece two tiles
//                 further is requested (two register sets: 64 KB per CU in flight)
//   barrier 2
// Status (round 4): correct everywhere it applies (1-3e-7 per frame against the oracle) and SLOWER than features_v2
// (1.70-1.77 ms against 1.13 ms on config 2), hence opt-in: at f32-grade accuracy a product costs three MFMAs plus the
// f16-pair hand-over on the vector pipe, and with two waves per SIMD nothing covers a wave's dependent chains (matrix pipes
// 34 % busy; DESIGN.md section 6d has the ablation table).
#include "features_dev.hpp"

// AF_V3_ABL (diagnostic builds only, scripts/ablate/): bit mask of work compiled OUT of the kernel to see what it costs.
// The results are then wrong; every guard that bounds a global access stays in.
//   1 stage-1 MFMAs   2 stage-1 conversion + Y stores   4 maxima + tile update   8 stage-2 MFMAs   16 stage-2 power + P stores
//   32 mel MFMAs   64 mel stores   128 PCM prefetch   256 mel operand reads   512 barriers 1..3   1024 the whole mel block
//   2048 the prefetch instructions themselves   4096 the stage-1 operand reads
#ifndef AF_V3_ABL
#define AF_V3_ABL 0
#endif
// experiment knob: 1 = waves 4..7 run stage 1 + tile update before the mel phase, waves 0..3 after it (measured: 1.79 ms
// against 1.75 in step, gpurun_out/r04/ablate_v3_4.txt)
#ifndef AF_V3_STAGGER
#define AF_V3_STAGGER 0
#endif

namespace af {
namespace feat {
namespace v3 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
using v2::UnitDesc;

// ---- PCM samples: one raw load per frame through a range-checked buffer resource (what lies behind the clip's end
// reads as 0, no branch), converted to mono where it is used --------------------------------------------------------
template <int CH, bool I16>
struct Smp;
template <>
struct Smp<2, false> {
    typedef uint2 raw_t;
    static constexpr int kBpf = 8;
    __device__ static __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, int off)
    {
        return __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    }
    __device__ static __forceinline__ float mono(raw_t r, int sel)
    {
        const float a = __uint_as_float(r.x), b = __uint_as_float(r.y);
        return sel < 0 ? (a + b) * 0.5f : sel ? b : a;          // to_mono: capture.rs:34-40
    }
};
template <>
struct Smp<1, false> {
    typedef unsigned raw_t;
    static constexpr int kBpf = 4;
    __device__ static __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, int off)
    {
        return __builtin_bit_cast(unsigned, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
    }
    __device__ static __forceinline__ float mono(raw_t r, int) { return __uint_as_float(r); }
};
template <>
struct Smp<2, true> {
    typedef unsigned raw_t;
    static constexpr int kBpf = 4;
    __device__ static __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, int off)
    {
        return __builtin_bit_cast(unsigned, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
    }
    __device__ static __forceinline__ float mono(raw_t r, int sel)
    {
        const float a = (float)(int16_t)(r & 0xffffu) * (1.0f / 32768.0f);
        const float b = (float)(int16_t)(r >> 16) * (1.0f / 32768.0f);
        return sel < 0 ? (a + b) * 0.5f : sel ? b : a;
    }
};
template <>
struct Smp<1, true> {
    typedef unsigned short raw_t;
    static constexpr int kBpf = 2;
    __device__ static __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, int off)
    {
        return __builtin_bit_cast(unsigned short, __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0));
    }
    __device__ static __forceinline__ float mono(raw_t r, int) { return (float)(int16_t)r * (1.0f / 32768.0f); }
};

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// max over the wave of a non-negative value, as its bit pattern (wave-uniform).  NaNs drop out (v_max_f32 returns the
// other operand).
__device__ __forceinline__ unsigned wave_max_bits(float x)
{
    x = fmaxf(x, dpp_f<0xB1>(x));          // quad_perm [1, 0, 3, 2]
    x = fmaxf(x, dpp_f<0x4E>(x));          // quad_perm [2, 3, 0, 1]
    x = fmaxf(x, dpp_f<0x141>(x));         // row_half_mirror
    x = fmaxf(x, dpp_f<0x140>(x));         // row_mirror: every lane of a row of 16 holds the row's maximum
    const int b = __float_as_int(x);
    const unsigned a0 = (unsigned)__builtin_amdgcn_readlane(b, 0), a1 = (unsigned)__builtin_amdgcn_readlane(b, 16);
    const unsigned a2 = (unsigned)__builtin_amdgcn_readlane(b, 32), a3 = (unsigned)__builtin_amdgcn_readlane(b, 48);
    const unsigned m01 = a0 > a1 ? a0 : a1, m23 = a2 > a3 ? a2 : a3;
    return m01 > m23 ? m01 : m23;
}

// |x| for the unit maximum: non-finite samples do not count (an Inf would push every other sample of the unit out of f16's
// range; the frames that contain it come out non-finite either way, the others stay exact)
__device__ __forceinline__ float finite_abs(float x)
{
    const float a = fabsf(x);
    return a < __builtin_inff() ? a : 0.0f;
}

// the unit's power-of-two scale from the bit pattern of max |x|:  s = 2^(140 - E) puts the maximum into [2^13, 2^14);
// c = 64 / s is what one factor of the power rows' scale is (X_acc = X s / 64)
struct Scale {
    float s, s2k, c;          // s, 2048 s, 64 / s
};
__device__ __forceinline__ Scale make_scale(unsigned max_bits)
{
    unsigned e = (max_bits >> 23) & 0xffu;
    e = e < 14u ? 14u : e > 254u ? 254u : e;
    Scale r;
    r.s = __uint_as_float((267u - e) << 23);
    r.s2k = __uint_as_float((278u - e) << 23);
    r.c = __uint_as_float((e - 7u) << 23);
    return r;
}

// four samples -> f16 pairs (hi, lo'), packed for one 8-byte store each
__device__ __forceinline__ void split4(const float (&m)[4], const Scale& sc, uint2& hi, uint2& lo)
{
    unsigned h[2], l[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float a = m[2 * q] * sc.s, b = m[2 * q + 1] * sc.s;
        const float a2 = m[2 * q] * sc.s2k, b2 = m[2 * q + 1] * sc.s2k;
        const f16x2 hv = __builtin_convertvector(f32x2{a, b}, f16x2);
        // 2048 (x - hi), exact: one v_fma_mix_f32 per value
        const float ra = __builtin_fmaf((float)hv[0], -2048.0f, a2), rb = __builtin_fmaf((float)hv[1], -2048.0f, b2);
        const f16x2 lv = __builtin_convertvector(f32x2{ra, rb}, f16x2);
        h[q] = __builtin_bit_cast(unsigned, hv);
        l[q] = __builtin_bit_cast(unsigned, lv);
    }
    hi = uint2{h[0], h[1]};
    lo = uint2{l[0], l[1]};
}

// tid through a register the compiler cannot see through: lane-dependent addresses derived from it are recomputed where
// they are used instead of being hoisted out of the unit loop (each hoisted one is a register held for the whole launch,
// and the kernel lives at its register limit)
__device__ __forceinline__ int opaque_tid(int tid)
{
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return tid + z;
}

typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ UnitDesc load_desc(const UnitDesc* units, unsigned u)
{
    // through the scalar cache, as a load the compiler tracks (features_v2.hip load_desc)
    typedef const __attribute__((address_space(4))) u32x8* cptr;
    const u32x8 r = reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(units))[__builtin_amdgcn_readfirstlane(u)];
    UnitDesc d;
    d.pcm0 = (int64_t)(((unsigned long long)r[1] << 32) | r[0]);
    d.out0 = (int64_t)(((unsigned long long)r[3] << 32) | r[2]);
    d.n_avail = (int)r[4];
    d.n_rows = (int)r[5];
    d.flags = (int)r[6];
    d.sel = (int)r[7];
    return d;
}

__device__ __forceinline__ f32x4 mfma16(const f16x8& a, const f16x8& b, const f32x4& c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// stage-2 product (ablation 8: no MFMA, the operands stay alive)
__device__ __forceinline__ f32x4 mfma16s2(const f16x8& a, const f16x8& b, const f32x4& c)
{
    if constexpr (AF_V3_ABL & 8) {
        const f32x4 bb = __builtin_bit_cast(f32x4, b);
        return f32x4{c[0] + bb[0], c[1] + bb[1], c[2] + bb[2], c[3] + bb[3]};
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
}
__device__ __forceinline__ f16x8 lds_frag(const unsigned char* p)
{
    return __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(p));
}

// Diagnostics (Params::stamps != null only): per-phase cycle sums of waves 0..3.
// lane-dependent indices, re-derived where they are used (see opaque_tid)
#define AF_LANE_VARS                                                                          \
    const int otid = opaque_tid(tid);                                                         \
    const int lane = otid & 63, f = lane & 15, g = lane >> 4, tn2 = otid & 31, tq = otid >> 5; \
    const int lane_pos = 128 * tq + tn2;                                                      \
    (void)lane; (void)f; (void)g; (void)tn2; (void)tq; (void)lane_pos

#define AF_STAMP3(i)                                                        \
    do {                                                                    \
        if constexpr (STAMPS) {                                             \
            __builtin_amdgcn_sched_barrier(0);                              \
            const unsigned long long _t = __builtin_amdgcn_s_memtime();     \
            __builtin_amdgcn_s_waitcnt(0xC07F);                             \
            acc_t[i] += _t - t_prev;                                        \
            t_prev = _t;                                                    \
            __builtin_amdgcn_sched_barrier(0);                              \
        }                                                                   \
    } while (0)

template <int CH, bool I16, bool SEP, int NSLOT, bool STAMPS, bool QUADS>
__global__ __launch_bounds__(kT, 2) void af_features_v3_kernel(const Params p)
{
    using S = Smp<CH, I16>;
    typedef typename S::raw_t raw_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Ybase = smem + kOffY;
    unsigned char* const Thi = smem + kOffThi;
    unsigned char* const Tlo = smem + kOffTlo;
    unsigned char* const Halo = smem + kOffHalo;
    float* const P = reinterpret_cast<float*>(smem + kOffP);
    unsigned* const wmax = reinterpret_cast<unsigned*>(smem + kOffWmax);
    unsigned* const wfresh = reinterpret_cast<unsigned*>(smem + kOffWfresh);
    unsigned char* const A2s = smem + kOffA2s;

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int64_t G = gridDim.x, bid = blockIdx.x;
    unsigned u = (unsigned)(p.total_units * bid / G);
    const unsigned u_end = (unsigned)(p.total_units * (bid + 1) / G);
    if (u >= u_end) return;

    // ---- constant fragments, resident for the whole launch: stage 1 (this wave's four n2), stage 2, this wave's mel slots
    f16x8 a1h[4][2], a1l[4][2], a2h[4], a2l[4];
    float wf[NSLOT * kSlotSteps];
    {
        const int lane = tid & 63;
        const uint4* t1 = p.tabs;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int n2 = 4 * w + t;
                a1h[t][rb] = __builtin_bit_cast(f16x8, t1[((n2 * 2 + rb) * 2 + 0) * 64 + lane]);
                a1l[t][rb] = __builtin_bit_cast(f16x8, t1[((n2 * 2 + rb) * 2 + 1) * 64 + lane]);
            }
        const uint4* t2 = p.tabs + kA1Vec;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            a2h[rb] = __builtin_bit_cast(f16x8, t2[(rb * 2 + 0) * 64 + lane]);
            a2l[rb] = __builtin_bit_cast(f16x8, t2[(rb * 2 + 1) * 64 + lane]);
        }
        const float4* wbase = reinterpret_cast<const float4*>(p.wfrag) + (size_t)w * kMaxSlotsW * 64;
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
            const float4 v = wbase[lane + q * 64];
            wf[4 * q] = v.x; wf[4 * q + 1] = v.y; wf[4 * q + 2] = v.z; wf[4 * q + 3] = v.w;
        }
        // the k1 = 16 matrix -> LDS (only the wave with the special item reads it)
        const uint4* t3 = p.tabs + kA1Vec + kA2Vec;
        if (tid < kA2sVec) reinterpret_cast<uint4*>(A2s)[tid] = t3[tid];
    }
    // first bin of every mel slot of this wave (wave-uniform)
    int lm[NSLOT];
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) lm[q] = __builtin_amdgcn_readfirstlane(p.meta[w * kMaxSlotsW + q]);
    // bins 513..519 stay zero (padding steps of a mel slot may touch them with zero weights)
    if (tid < 16 * 7) P[513 * 16 + tid] = 0.0f;
    // a wait the compiler sees for the loads above (features_v2.hip: loop-invariant loads carried as "pending")
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) asm volatile("" : "+v"(a1h[t][rb]), "+v"(a1l[t][rb]));
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) asm volatile("" : "+v"(a2h[rb]), "+v"(a2l[rb]));
#pragma unroll
    for (int q = 0; q < NSLOT * kSlotSteps; ++q) asm volatile("" : "+v"(wf[q]));

    // this wave's share of the mel plan
    int slot_end_w[NSLOT];
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) slot_end_w[q] = p.plan.slot_end[w][q];
    const int n_mels = p.plan.n_mels;

    // lane constants of the tile update: sample pair p = tid + 512 r  ->  n2 = p & 31, q = p >> 5 (r = 0; + 16 per r).
    // Tile position of sample j of quad Q: 32 (4 Q + j) + n2.

    // PCM frames [frame0, frame0 + n) as a buffer: n <= 0 gives an empty range (every load returns 0, nothing is fetched)
    // (every component through readfirstlane: the unit record may live in vector registers, and a resource the compiler
    // cannot prove uniform turns each load into a waterfall loop -- behind which it waits for vmcnt(0) everywhere)
    const auto clip_rsrc = [&](int64_t frame0, int n) {
        n = n < 0 ? 0 : n > (1 << 20) ? (1 << 20) : n;
        const unsigned long long a = reinterpret_cast<unsigned long long>(p.pcm) + (unsigned long long)frame0 * S::kBpf;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(n * S::kBpf), 0x00020000);
    };
    // the new samples of tile (unit of d + k): tile positions [768, 4864) = PCM frames d.pcm0 + 4096 k + position
    const auto issue_piece = [&](raw_t (&dst)[2][4], const UnitDesc& d, int k, bool need) {
        const __amdgpu_buffer_rsrc_t rs = clip_rsrc(d.pcm0 + 4096 * k, (need && !(AF_V3_ABL & 128)) ? d.n_avail - 4096 * k : 0);
        AF_LANE_VARS;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[r][j] = S::load(rs, (lane_pos + 768 + 2048 * r + 32 * j) * S::kBpf);
    };

    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = STAMPS ? __builtin_amdgcn_s_memtime() : 0;

    // The issue arbiter prefers the older wave of a SIMD: waves 4..7 (second half of the workgroup) would lose every
    // contended slot to waves 0..3.  AF_V3_PRIO = 1 (experiment knob) gives them one priority step: measured level (1.77
    // against 1.75 ms), off.
#ifndef AF_V3_PRIO
#define AF_V3_PRIO 0
#endif
    if (AF_V3_PRIO && w >= 4) __builtin_amdgcn_s_setprio(AF_V3_PRIO);

    bool fresh = true;
    UnitDesc ud_next = load_desc(p.units, u);
    // the unit whose power rows sit in P: its mel phase runs at the top of the NEXT pass, beside that pass's stage 1
    int64_t prev_out0 = 0;
    int prev_rows = 0;
    float prev_c = 1.0f;
    float c_cur = 1.0f;            // 64 / s of the tile stage 1 reads next
    unsigned tail_prev = 0;        // bits of max |x| over that tile's last 768 samples
    int hb = 0, tb = 0;            // halo buffer holding those samples as f32; signal image stage 1 reads next
    float mn[2][4];                // mono samples of the piece that completes the NEXT tile (their maxima are in wmax)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) mn[r][j] = 0.0f;
    // Two prefetch sets: at pass u the piece of tile u + 2 is consumed, the piece of tile u + 3 stays in flight and the
    // piece of tile u + 4 is requested (one workgroup per CU: a unit lasts about 2 us, a loaded HBM round trip up to 4).
    // The loop body exists twice, once per set.
    raw_t pf[2][2][4];

    // mel of the unit described by (out0, rows, c): banded W (A, registers) x P^T (B, bins x 16 frames from the bin-major
    // power rows) on v_mfma_f32_16x16x4_f32.  Every wave issues exactly NSLOT (QUADS) or 4 NSLOT store instructions (rows
    // behind `rows` are out of the buffer's range, what must not be written gets an out-of-range offset): vector memory
    // operations retire in order, the wait for a prefetched piece is "all but the youngest N", and with a count the
    // compiler cannot see it waits for the stores issued just before it.  (log-mel output stays with features_v2.)
    // Branch-free on purpose (features_kernel.hip, pitfall (iii): a wave-uniform branch in an MFMA stream costs about 150
    // cycles of instruction fetch; with three of them per slot the phase took 3 500 cycles for 1 300 cycles of matrix work):
    // every compiled-in slot runs (slots a wave does not use carry zero weights and no group end), a group end is a select.
    const auto mel_phase = [&](int64_t out0, int rows, float cc) {
        if constexpr (!(AF_V3_ABL & 1024)) {
            AF_LANE_VARS;
            const float* pbase = P + 16 * g + f;             // PT[bin = kq][frame]
            const unsigned row_off = (unsigned)f * (unsigned)n_mels + 4u * (unsigned)g;
            float bop[NSLOT * kSlotSteps];
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) {
                const float* bq = pbase + 16 * lm[q];    // step i of the slot: bins bin0 + 4 i + kq
#pragma unroll
                for (int i = 0; i < kSlotSteps; ++i) bop[4 * q + i] = (AF_V3_ABL & 256) ? (float)(lane + i + q) : bq[64 * i];
            }
            const int rows_bytes = __builtin_amdgcn_readfirstlane(rows * n_mels * 4);
            const unsigned long long oa = reinterpret_cast<unsigned long long>(p.out) + (unsigned long long)(out0 * n_mels) * 4ull;
            const unsigned olo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)oa);
            const unsigned ohi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(oa >> 32));
            const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void*>(((unsigned long long)ohi << 32) | olo), 0, (AF_V3_ABL & 64) ? 0 : rows_bytes, 0x00020000);
            const int i0 = 4 * g;
            constexpr unsigned kOob = 0x7fffffffu;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) {
#pragma unroll
                for (int st = q * kSlotSteps; st < (q + 1) * kSlotSteps; st += 2) {
                    if constexpr (!(AF_V3_ABL & 32)) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[st], bop[st], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[st + 1], bop[st + 1], acc1, 0, 0, 0);
                    } else {
                        acc0[0] += wf[st] * bop[st];
                        acc1[0] += wf[st + 1] * bop[st + 1];
                    }
                }
                const int e = slot_end_w[q];             // wave-uniform (kernarg); 0: the group continues
                const int mm = e & 255, cnt = (e >> 8) & 31;
                const float keep = e != 0 ? 0.0f : 1.0f; // the accumulators restart behind a group end
                f32x4 r = acc0 + acc1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    r[i] = (r[i] * cc) * cc;             // the tile's scale leaves here
                    acc0[i] *= keep;
                    acc1[i] *= keep;
                }
                const unsigned off = (row_off + (unsigned)mm) * 4u;
                if constexpr (QUADS) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r), ors, (int)((e != 0 && i0 < cnt) ? off : kOob), 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r[i]), ors, (int)((e != 0 && i0 + i < cnt) ? off + 4u * i : kOob), 0, 0);
                }
            }
        }
    };

    const auto unit = [&](auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        UnitDesc ud = ud_next;
        ud_next = load_desc(p.units, u + 1);             // (the table ends with spare records)
        // (a workgroup with an odd number of units runs one more pass, on a record that stores nothing: see the loop below)
        const bool live = u < u_end;
        if (!live) { ud.n_rows = 0; ud.n_avail = 0; ud.flags = 0; }
        const int sel = SEP ? ud.sel : -1;
        const int run = (ud.flags >> 8) & 255;           // units that follow in this clip
        const bool has_next = (u + 1 < u_end) && run >= 1;
        AF_STAMP3(0);

        if (__builtin_expect(fresh, 0)) {
            // ---- a clip starts: the whole tile AND the piece that completes the next one, straight from global memory ----
            __syncthreads();                             // nobody still reads T, the halo or the maxima
            AF_LANE_VARS;
            // (one after the other, each converted before the next is requested: a clip start may wait, it must not be the
            // place that sets the kernel's register count)
            float m[3][4];
            float mx = 0.0f, mt = 0.0f;
            {
                const __amdgpu_buffer_rsrc_t rs = clip_rsrc(ud.pcm0, ud.n_avail);
                raw_t raw[3][4];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) raw[r][j] = S::load(rs, (lane_pos + 2048 * r + 32 * j) * S::kBpf);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // (r = 2: quads 32..37 exist for tq < 6 only; the others read past the tile and are not used)
                        m[r][j] = S::mono(raw[r][j], sel);
                        if (r == 2 && tq >= kQuads - 32) m[r][j] = 0.0f;
                        mx = fmaxf(mx, finite_abs(m[r][j]));
                        if (r == 2) mt = fmaxf(mt, finite_abs(m[r][j]));        // the last 768 samples
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const unsigned wm = wave_max_bits(mx), wt = wave_max_bits(mt);
                if (lane == 0) { wfresh[2 * w] = wm; wfresh[2 * w + 1] = wt; }
            }
            // the next tile's piece: mono samples and maxima, as the steady state leaves them behind
            {
                raw_t nu1[2][4];
                issue_piece(nu1, ud, 1, has_next);
                float nx = 0.0f, nt = 0.0f;
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        mn[r][j] = S::mono(nu1[r][j], sel);
                        nx = fmaxf(nx, finite_abs(mn[r][j]));
                        if (r == 1) nt = fmaxf(nt, finite_abs(mn[r][j]));
                    }
                const unsigned wm = wave_max_bits(nx);
                const unsigned wt = w >= 5 ? wave_max_bits(nt) : 0u;
                if (lane == 0) { wmax[2 * w] = wm; wmax[2 * w + 1] = wt; }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the pieces of tiles u + 2 and u + 3 go out now, each into its set -- with as many (empty-range) stores between
            // them as a pass issues, so that the first has the same number of younger operations behind it as in the steady state
            issue_piece(pf[PAR], ud, 2, (u + 2 < u_end) && run >= 2);
            {
                const __amdgpu_buffer_rsrc_t nowhere = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0, 0x00020000);
#pragma unroll
                for (int q = 0; q < (QUADS ? NSLOT : 4 * NSLOT); ++q) {
                    // (distinct offsets: identical stores would be merged into one)
                    if constexpr (QUADS) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, nowhere, 0x7ffff000 + 16 * q, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b32(0u, nowhere, 0x7ffff000 + 16 * q, 0, 0);
                }
            }
            issue_piece(pf[PAR ^ 1], ud, 3, (u + 3 < u_end) && run >= 3);
            __syncthreads();
            unsigned mall = 0, mtl = 0;
#pragma unroll
            for (int ww = 0; ww < kW; ww += 2) {
                const uint4 v = reinterpret_cast<const uint4*>(wfresh)[ww / 2];
                mall = v.x > mall ? v.x : mall;
                mall = v.z > mall ? v.z : mall;
                mtl = v.y > mtl ? v.y : mtl;
                mtl = v.w > mtl ? v.w : mtl;
            }
            const Scale sc = make_scale(mall);
            c_cur = sc.c;
            tail_prev = mtl;
            hb = 0;
            tb = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int quad = tq + 16 * r;
                if (quad < kQuads) {
                    uint2 hi, lo;
                    split4(m[r], sc, hi, lo);
                    *reinterpret_cast<uint2*>(Thi + tn2 * kTRow + quad * 8) = hi;
                    *reinterpret_cast<uint2*>(Tlo + tn2 * kTRow + quad * 8) = lo;
                    if (r == 2)
                        *reinterpret_cast<float4*>(Halo + tn2 * kHaloRow + (quad - 32) * 16) =
                            float4{m[r][0], m[r][1], m[r][2], m[r][3]};
                }
            }
            __syncthreads();                             // T ready
        }
        AF_STAMP3(1);

        // ======== first half of the pass: mel of the previous unit, stage 1 of this one, tile update for the next ========
        // (AF_V3_STAGGER: the two waves of a SIMD, w and w + 4, take the two jobs in opposite order -- f32 matrix work
        // beside vector-heavy work instead of both queueing for the same pipe; measured slower, off)
        float c_next = c_cur;
        unsigned tail_next = tail_prev;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
        if ((half == 0) == (w < 4 || !AF_V3_STAGGER)) {
        mel_phase(prev_out0, prev_rows, prev_c);
        AF_STAMP3(2);
        } else {

        // ---- stage 1: this wave's four n2, two at a time (32 accumulator registers live) ----
        __builtin_amdgcn_sched_barrier(0);
        {
            AF_LANE_VARS;
            const unsigned char* th = Thi + tb * kTBuf;
            const unsigned char* tl = Tlo + tb * kTBuf;
            const int bro = 16 * (f + g);
            // -> f16 pairs -> Y[k1 = 4g + i][f][row block rb: Yr | Yi][hi | lo'][n2 = 4w .. 4w+3]:  hi = f16(P1),
            // lo' = f16(2048 (P1 - hi) + P2)  (the value is P1 + P2 / 2048; |lo'| <= |P1| + |P2| < 2^14 with the 1/64 in A1)
            unsigned hw[2][2][4], lw[2][2][4];           // [n2 pair][rb][i]: packed (n2 even, n2 odd)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 p1[2][2], p2[2][2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int t = 2 * q + tt, n2 = 4 * w + t;
                    const f16x8 bh = lds_frag(th + n2 * kTRow + bro);
                    const f16x8 bl = lds_frag(tl + n2 * kTRow + bro);
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        if constexpr (!(AF_V3_ABL & 1)) {
                            p1[tt][rb] = mfma16(a1h[t][rb], bh, z);
                            p2[tt][rb] = mfma16(a1h[t][rb], bl, z);
                            p2[tt][rb] = mfma16(a1l[t][rb], bh, p2[tt][rb]);
                        } else {
                            p1[tt][rb] = __builtin_bit_cast(f32x4, __builtin_bit_cast(uint4, bh));
                            p2[tt][rb] = __builtin_bit_cast(f32x4, __builtin_bit_cast(uint4, bl));
                        }
                    }
                }
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float a = p1[0][rb][i], b = p1[1][rb][i];
                        const f16x2 hv = __builtin_convertvector(f32x2{a, b}, f16x2);
                        const float a2 = __builtin_fmaf(a, 2048.0f, p2[0][rb][i]), b2 = __builtin_fmaf(b, 2048.0f, p2[1][rb][i]);
                        const float ra = __builtin_fmaf((float)hv[0], -2048.0f, a2);
                        const float rbb = __builtin_fmaf((float)hv[1], -2048.0f, b2);
                        const f16x2 lv = __builtin_convertvector(f32x2{ra, rbb}, f16x2);
                        hw[q][rb][i] = __builtin_bit_cast(unsigned, hv);
                        lw[q][rb][i] = __builtin_bit_cast(unsigned, lv);
                    }
                if (q == 0) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (AF_V3_ABL & 2) {
                unsigned sink = 0;
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) sink += hw[q][rb][0] + hw[q][rb][1] + lw[q][rb][2] + lw[q][rb][3];
                if (sink == 0x12345678u) wmax[lane] = 1u;
            } else {
                // (the 16-byte piece n2 / 8 = w >> 1 of a part sits at position (w >> 1) ^ (f >> 2))
                unsigned char* yw = Ybase + (4 * g) * kYSlot + f * kYRow + 16 * ((w >> 1) ^ (f >> 2)) + 8 * (w & 1);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        *reinterpret_cast<uint2*>(yw + i * kYSlot + rb * 64) = uint2{hw[0][rb][i], hw[1][rb][i]};
                        *reinterpret_cast<uint2*>(yw + i * kYSlot + rb * 64 + 128) = uint2{lw[0][rb][i], lw[1][rb][i]};
                    }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        AF_STAMP3(3);

        // ---- tile update: the next unit's tile = this one's last 768 samples (f32 copy in the halo buffer) + the piece in
        // mn, into the OTHER signal image (this one is being read by the slower waves' stage 1) ----
        if (__builtin_expect(has_next && !(AF_V3_ABL & 4), 1)) {
            AF_LANE_VARS;
            unsigned mall = tail_prev, mtl = 0;
#pragma unroll
            for (int ww = 0; ww < kW; ww += 2) {
                const uint4 v = reinterpret_cast<const uint4*>(wmax)[ww / 2];
                mall = v.x > mall ? v.x : mall;
                mall = v.z > mall ? v.z : mall;
                mtl = v.y > mtl ? v.y : mtl;
                mtl = v.w > mtl ? v.w : mtl;
            }
            const Scale sc = make_scale(mall);
            c_next = sc.c;
            tail_next = mtl;
            unsigned char* th = Thi + (tb ^ 1) * kTBuf;
            unsigned char* tl = Tlo + (tb ^ 1) * kTBuf;
            if (w < 3) {                                 // quads 0..5 <- halo (threads 0..191)
                const float4 hv = *reinterpret_cast<const float4*>(Halo + hb * (32 * kHaloRow) + tn2 * kHaloRow + tq * 16);
                const float hm[4] = {hv.x, hv.y, hv.z, hv.w};
                uint2 hi, lo;
                split4(hm, sc, hi, lo);
                *reinterpret_cast<uint2*>(th + tn2 * kTRow + tq * 8) = hi;
                *reinterpret_cast<uint2*>(tl + tn2 * kTRow + tq * 8) = lo;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int quad = 6 + tq + 16 * r;
                uint2 hi, lo;
                split4(mn[r], sc, hi, lo);
                *reinterpret_cast<uint2*>(th + tn2 * kTRow + quad * 8) = hi;
                *reinterpret_cast<uint2*>(tl + tn2 * kTRow + quad * 8) = lo;
                if (r == 1 && w >= 5)
                    *reinterpret_cast<float4*>(Halo + (hb ^ 1) * (32 * kHaloRow) + tn2 * kHaloRow + (quad - 32) * 16) =
                        float4{mn[r][0], mn[r][1], mn[r][2], mn[r][3]};
            }
        }
        AF_STAMP3(5);
        }
        }
        if constexpr (!(AF_V3_ABL & 512)) __syncthreads();   // B1: Y ready; the next signal image and halo written; maxima read
        AF_STAMP3(4);

        // ======== second half: stage 2 (two k1 per wave -> power rows), the piece of tile u + 2 becomes mn ========
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            AF_LANE_VARS;
            const int slot = (2 * w + 1 + it) & 15;      // k1; 0: the special item (k1 = 0 and k1 = 16)
            const unsigned char* yb = Ybase + slot * kYSlot + f * kYRow + 16 * (g ^ (f >> 2));
            const f16x8 yrh = lds_frag(yb), yih = lds_frag(yb + 64), yrl = lds_frag(yb + 128), yil = lds_frag(yb + 192);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            // row 4g + i of row block rb is k2 = 8 rb + 4 (g >> 1) + (g & 1) + 2 (i >> 1), cos (i even) or -sin (i odd):
            // the two lane groups of a 32-lane store half differ in the parity of k2, i.e. by 32 banks
            const int k2g = 4 * (g >> 1) + (g & 1);
            float* prow = P + f;                         // PT[bin][f]: bin b is prow[16 b]
            if (__builtin_expect(slot != 0, 1)) {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    const f32x4 a0p1 = mfma16s2(a2h[rb], yrh, z);
                    const f32x4 a1p1 = mfma16s2(a2h[rb], yih, z);
                    f32x4 a0p2 = mfma16s2(a2h[rb], yrl, z);
                    f32x4 a1p2 = mfma16s2(a2h[rb], yil, z);
                    a0p2 = mfma16s2(a2l[rb], yrh, a0p2);
                    a1p2 = mfma16s2(a2l[rb], yih, a1p2);
                    float pw[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float re = __builtin_fmaf(a0p2[2 * h] - a1p2[2 * h + 1], 1.0f / 2048.0f, a0p1[2 * h] - a1p1[2 * h + 1]);
                        const float im = __builtin_fmaf(a1p2[2 * h] + a0p2[2 * h + 1], 1.0f / 2048.0f, a1p1[2 * h] + a0p1[2 * h + 1]);
                        pw[h] = __builtin_fmaf(re, re, im * im);
                    }
                    const int k2 = 8 * rb + k2g;         // and k2 + 2
                    if constexpr (!(AF_V3_ABL & 16)) {
                        if (rb < 2) {
                            prow[16 * (slot + 32 * k2)] = pw[0];
                            prow[16 * (slot + 32 * k2 + 64)] = pw[1];
                        } else {                         // bin 1024 - k
                            prow[16 * (1024 - slot - 32 * k2)] = pw[0];
                            prow[16 * (1024 - slot - 32 * k2 - 64)] = pw[1];
                        }
                    }
                }
            } else {
                // k1 = 0: X[32 k2] = A2 Y0 (real input), k2 = 0..16
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) {
                    const f32x4 p1 = mfma16s2(a2h[rb], yrh, z);
                    f32x4 p2 = mfma16s2(a2h[rb], yrl, z);
                    p2 = mfma16s2(a2l[rb], yrh, p2);
                    float pw[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float re = __builtin_fmaf(p2[2 * h], 1.0f / 2048.0f, p1[2 * h]);
                        const float im = __builtin_fmaf(p2[2 * h + 1], 1.0f / 2048.0f, p1[2 * h + 1]);
                        pw[h] = __builtin_fmaf(re, re, im * im);
                    }
                    const int k2 = 8 * rb + k2g;
                    if constexpr (!(AF_V3_ABL & 16)) {
                        if (rb < 2) {
                            prow[16 * (32 * k2)] = pw[0];
                            prow[16 * (32 * k2 + 64)] = pw[1];
                        } else if (g == 0) {
                            prow[16 * 512] = pw[0];      // k2 = 16: bin 512
                        }
                    }
                }
                // k1 = 16: X[16 + 32 k2] = A2s Y16, k2 = 0..15 (the matrix comes from LDS: one wave uses it)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const f16x8 sh = lds_frag(A2s + ((rb * 2 + 0) * 64 + lane) * 16);
                    const f16x8 sl = lds_frag(A2s + ((rb * 2 + 1) * 64 + lane) * 16);
                    const f32x4 p1 = mfma16s2(sh, yih, z);
                    f32x4 p2 = mfma16s2(sh, yil, z);
                    p2 = mfma16s2(sl, yih, p2);
                    float pw[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float re = __builtin_fmaf(p2[2 * h], 1.0f / 2048.0f, p1[2 * h]);
                        const float im = __builtin_fmaf(p2[2 * h + 1], 1.0f / 2048.0f, p1[2 * h + 1]);
                        pw[h] = __builtin_fmaf(re, re, im * im);
                    }
                    const int k2 = 8 * rb + k2g;
                    if constexpr (!(AF_V3_ABL & 16)) {
                        prow[16 * (16 + 32 * k2)] = pw[0];
                        prow[16 * (16 + 32 * k2 + 64)] = pw[1];
                    }
                }
            }
        }
        AF_STAMP3(6);

        // ---- the piece of tile u + 2 (set PAR, requested two passes ago): mono samples and their maxima for the next
        // pass's tile update (the maxima of the previous piece were read before barrier 1) ----
        if (__builtin_expect((u + 2 < u_end) && run >= 2 && !(AF_V3_ABL & 4), 1)) {
            const int lane = opaque_tid(tid) & 63;
            float mx = 0.0f, mt = 0.0f;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mn[r][j] = S::mono(pf[PAR][r][j], sel);
                    mx = fmaxf(mx, finite_abs(mn[r][j]));
                    if (r == 1) mt = fmaxf(mt, finite_abs(mn[r][j]));
                }
            const unsigned wm = wave_max_bits(mx);
            // quads 32..37 (that tile's last 768 samples) are r = 1 of threads 320..511 = waves 5..7
            const unsigned wt = w >= 5 ? wave_max_bits(mt) : 0u;
            if (lane == 0) { wmax[2 * w] = wm; wmax[2 * w + 1] = wt; }
        }
        // ---- the piece of tile u + 4 goes into the set just consumed: the LAST vector memory operations of the pass and
        // unconditional (an empty range when there is nothing to fetch), so that "all but the youngest N" is what the next
        // consumer has to wait for on every path ----
        if constexpr (!(AF_V3_ABL & 2048)) issue_piece(pf[PAR], ud, 4, (u + 4 < u_end) && run >= 4);
        AF_STAMP3(7);
        if constexpr (!(AF_V3_ABL & 512)) __syncthreads();   // B2: power rows ready; Y free; maxima published
        AF_STAMP3(4);

        prev_out0 = ud.out0;
        prev_rows = ud.n_rows;
        prev_c = c_cur;
        c_cur = c_next;
        tail_prev = tail_next;
        if (has_next) { hb ^= 1; tb ^= 1; }
        fresh = !has_next;
        ++u;
    };

    // Two passes per trip, no exit between them: a `break` in the middle leaves a (never taken) edge in the control-flow
    // graph on which the set issued at the end of the first pass is the youngest thing in flight when that pass starts
    // again, and the compiler then waits for vmcnt(0) at every consumption.  An odd unit count is padded with one dead pass.
    while (u < u_end) {
        unit(std::integral_constant<int, 0>{});
        unit(std::integral_constant<int, 1>{});
    }
    mel_phase(prev_out0, prev_rows, prev_c);             // the last unit's rows
    if constexpr (STAMPS) {
        if ((tid & 63) == 0 && w >= p.stamp_wave0 && w < p.stamp_wave0 + 4)
            for (int i = 0; i < 8; ++i) atomicAdd(p.stamps + (w - p.stamp_wave0) * 8 + i, acc_t[i]);
    }
}

template <int CH, bool I16, bool SEP, int NSLOT, bool STAMPS, bool QUADS>
static af_status launch_k(const Params& p, int grid, hipStream_t stream)
{
    auto kern = af_features_v3_kernel<CH, I16, SEP, NSLOT, STAMPS, QUADS>;
    static LdsAttrFlags attr_set;
    AF_HIP(allow_big_lds(kern, kLdsBytes, attr_set));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kT), kLdsBytes, stream, p);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

template <int CH, bool I16>
static af_status launch_fmt(const Params& p, bool separate, int grid, hipStream_t stream)
{
    int most = 0;
    // QUADS: every group of the plan starts and ends on a multiple of four filters and the rows are 16-byte aligned:
    // one 16-byte store per slot (16-byte buffer stores need 16-byte-aligned addresses); otherwise four 4-byte stores
    bool quads = p.plan.n_mels % 4 == 0 && reinterpret_cast<uintptr_t>(p.out) % 16 == 0;
    for (int w = 0; w < kW; ++w) {
        most = p.plan.wave_nslots[w] > most ? p.plan.wave_nslots[w] : most;
        for (int q = 0; q < kMaxSlotsW; ++q) {
            const int e = p.plan.slot_end[w][q];
            if (e != 0 && (((e & 255) | ((e >> 8) & 31)) & 3) != 0) quads = false;
        }
    }
    if (most > 8) return fail(AF_ERR_UNSUPPORTED, "features v3: %d mel slots on one wave (limit 8)", most);
    // diagnostics: the one stamping instance is stereo f32, downmix, a quad-aligned plan of at most 7 slots per wave --
    // anything else would run an instance without stamps and hand back zeros as if they were measurements
    if (p.stamps && !(CH == 2 && !I16 && !separate && quads && most <= 7))
        return fail(AF_ERR_UNSUPPORTED, "features v3: in-kernel stamps exist for stereo f32 downmix with a quad-aligned mel plan of <= 7 slots per wave only");
    if constexpr (CH == 2) {
        if (separate) return quads ? launch_k<CH, I16, true, 8, false, true>(p, grid, stream) : launch_k<CH, I16, true, 8, false, false>(p, grid, stream);
    }
    if constexpr (CH == 2 && !I16) {
        if (p.stamps && quads && most <= 7) return launch_k<CH, I16, false, 7, true, true>(p, grid, stream);
    }
    if (!quads) return launch_k<CH, I16, false, 8, false, false>(p, grid, stream);
    if (most <= 6) return launch_k<CH, I16, false, 6, false, true>(p, grid, stream);
    if (most <= 7) return launch_k<CH, I16, false, 7, false, true>(p, grid, stream);
    return launch_k<CH, I16, false, 8, false, true>(p, grid, stream);
}

af_status launch(const Params& p, int channels, bool i16, bool separate, int grid, hipStream_t stream)
{
    if (grid <= 0) return AF_OK;
    if (channels == 2 && !i16) return launch_fmt<2, false>(p, separate, grid, stream);
    if (channels == 1 && !i16) return launch_fmt<1, false>(p, false, grid, stream);
    if (channels == 2 && i16) return launch_fmt<2, true>(p, separate, grid, stream);
    if (channels == 1 && i16) return launch_fmt<1, true>(p, false, grid, stream);
    return fail(AF_ERR_UNSUPPORTED, "features: channels must be 1 or 2 (got %d)", channels);
}

// ---- host: fragment tables -------------------------------------------------------------------------------------------
namespace {
constexpr double kPi = 3.14159265358979323846;

// x -> f16 pair (hi, lo' = f16((x - hi) 2^11)) as bit patterns
void split_host(double x, uint16_t& hi, uint16_t& lo)
{
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)((x - (double)h) * 2048.0);
    hi = __builtin_bit_cast(uint16_t, h);
    lo = __builtin_bit_cast(uint16_t, l);
}

// fragment of a matrix A[row][k] (row block rb, 16 rows; K = 32) for lane l: row l & 15, k = 8 (l >> 4) + j
template <class F>
void put_frag(uint4* hi_dst, uint4* lo_dst, F&& a)
{
    for (int l = 0; l < 64; ++l) {
        uint16_t h[8], q[8];
        for (int j = 0; j < 8; ++j) split_host(a(l & 15, 8 * (l >> 4) + j), h[j], q[j]);
        hi_dst[l] = uint4{(unsigned)h[0] | (unsigned)h[1] << 16, (unsigned)h[2] | (unsigned)h[3] << 16,
                          (unsigned)h[4] | (unsigned)h[5] << 16, (unsigned)h[6] | (unsigned)h[7] << 16};
        lo_dst[l] = uint4{(unsigned)q[0] | (unsigned)q[1] << 16, (unsigned)q[2] | (unsigned)q[3] << 16,
                          (unsigned)q[4] | (unsigned)q[5] << 16, (unsigned)q[6] | (unsigned)q[7] << 16};
    }
}
}  // namespace

void build_tables(const double* win, uint4* tabs)
{
    // a1 [n2][rb][hi, lo'][64]: rows of row block 0 = Re k1 = 0..15; row block 1 = row 0: Re k1 = 16 WITHOUT the
    // twiddle W1024^(16 n2) (it sits in a2s), rows 1..15: Im k1 = 1..15.  Scaled by 1/64 so that |Y'| <= 2^12 for a
    // tile whose maximum is below 2^14 (sum over n1 of the Hann window = 16): the lo' word of a Y' value is bounded by
    // |P1| + |P2| <= 3 x 2^12.
    for (int n2 = 0; n2 < 32; ++n2)
        for (int rb = 0; rb < 2; ++rb) {
            uint4* dst = tabs + ((n2 * 2 + rb) * 2) * 64;
            put_frag(dst, dst + 64, [&](int m, int n1) {
                const int n = 32 * n1 + n2;
                const double wv = win[n] / 64.0;
                if (rb == 0) return wv * std::cos(2.0 * kPi * (double)((n * m) % 1024) / 1024.0);
                if (m == 0) return (n1 & 1) ? -wv : wv;
                return -wv * std::sin(2.0 * kPi * (double)((n * m) % 1024) / 1024.0);
            });
        }
    // a2 [rb][hi, lo'][64]: row 4 q + i of row block rb is k2 = 8 rb + 4 (q >> 1) + (q & 1) + 2 (i >> 1); i even: cos,
    // i odd: -sin of 2 pi n2 k2 / 32
    const auto k2_of = [](int rb, int m) { const int q = m >> 2, i = m & 3; return 8 * rb + 4 * (q >> 1) + (q & 1) + 2 * (i >> 1); };
    uint4* t2 = tabs + kA1Vec;
    for (int rb = 0; rb < 4; ++rb)
        put_frag(t2 + (rb * 2) * 64, t2 + (rb * 2 + 1) * 64, [&](int m, int n2) {
            const double a = 2.0 * kPi * (double)((n2 * k2_of(rb, m)) % 32) / 32.0;
            return (m & 1) ? -std::sin(a) : std::cos(a);
        });
    // a2s (k1 = 16): cos / -sin of 2 pi n2 (2 k2 + 1) / 64
    uint4* t3 = tabs + kA1Vec + kA2Vec;
    for (int rb = 0; rb < 2; ++rb)
        put_frag(t3 + (rb * 2) * 64, t3 + (rb * 2 + 1) * 64, [&](int m, int n2) {
            const double a = 2.0 * kPi * (double)((n2 * (2 * k2_of(rb, m) + 1)) % 64) / 64.0;
            return (m & 1) ? -std::sin(a) : std::cos(a);
        });
}

}  // namespace v3
}  // namespace feat
}  // namespace af
