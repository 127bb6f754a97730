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
// n2, 64 VGPRs, and the stage-2 fragments, 32 + 16, stay resident), a persistent grid over contiguous unit ranges.
//   stage 1   wave w: n2 = 4w .. 4w+3; B operand = 16 bytes of the f16 signal image T[n2][blk] (hi and lo' planes; frames
//             overlap, so lanes (f, g) with equal f + g read the same address); results -> f16 pairs -> Y in LDS,
//             [k1][f][Yr hi | Yi hi | Yr lo' | Yi lo'] (one ds_write_b64 = four n2)
//   barrier 1
//   tile update for the next unit (the prefetched PCM -> mono -> scaled f16 pairs -> T), then
//   stage 2   wave w: k1 = 2w+1, 2w+2 (wave 7: k1 = 15 and the special item) -> power rows P[f][bin] in LDS
//   barrier 2
//   mel       the banded f32 contraction of features_v2.hip on v_mfma_f32_16x16x4_f32 (same plan, same fragments)
#include "features_dev.hpp"

namespace af {
namespace feat {
namespace v3 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
using v2::UnitDesc;

// ---- PCM samples: one raw load per frame, converted to mono where it is used ------------------------------------------
template <int CH, bool I16>
struct Smp;
template <>
struct Smp<2, false> {
    typedef uint2 raw_t;
    static constexpr int kBpf = 8;
    __device__ static __forceinline__ raw_t zero() { return uint2{0u, 0u}; }
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
    __device__ static __forceinline__ raw_t zero() { return 0u; }
    __device__ static __forceinline__ float mono(raw_t r, int) { return __uint_as_float(r); }
};
template <>
struct Smp<2, true> {
    typedef unsigned raw_t;
    static constexpr int kBpf = 4;
    __device__ static __forceinline__ raw_t zero() { return 0u; }
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
    __device__ static __forceinline__ raw_t zero() { return 0; }
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

// the unit's power-of-two scale from the bit pattern of max |x|:  s = 2^(140 - E) puts the maximum into [2^13, 2^14);
// c = 16 / s is what one factor of the power rows' scale is (X_acc = X s / 16)
struct Scale {
    float s, s2k, c;          // s, 2048 s, 16 / s
};
__device__ __forceinline__ Scale make_scale(unsigned max_bits)
{
    unsigned e = (max_bits >> 23) & 0xffu;
    e = e < 14u ? 14u : e > 254u ? 254u : e;
    Scale r;
    r.s = __uint_as_float((267u - e) << 23);
    r.s2k = __uint_as_float((278u - e) << 23);
    r.c = __uint_as_float((e - 9u) << 23);
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

typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
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
__device__ __forceinline__ f16x8 lds_frag(const unsigned char* p)
{
    return __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(p));
}

template <int CH, bool I16, bool SEP, int NSLOT>
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
    int* const lmeta = reinterpret_cast<int*>(smem + kOffMeta);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = lane & 15, g = lane >> 4;

    const int64_t G = gridDim.x, bid = blockIdx.x;
    unsigned u = (unsigned)(p.total_units * bid / G);
    const unsigned u_end = (unsigned)(p.total_units * (bid + 1) / G);
    if (u >= u_end) return;

    // ---- constant fragments, resident for the whole launch ----
    f16x8 a1h[4][2], a1l[4][2], a2h[4], a2l[4], a2sh[2], a2sl[2];
    {
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
        const uint4* t3 = p.tabs + kA1Vec + kA2Vec;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            a2sh[rb] = __builtin_bit_cast(f16x8, t3[(rb * 2 + 0) * 64 + lane]);
            a2sl[rb] = __builtin_bit_cast(f16x8, t3[(rb * 2 + 1) * 64 + lane]);
        }
    }
    for (int i = tid; i < kW * kMaxSlotsW; i += kT) lmeta[i] = p.meta[i];
    // bins 513..519 of every power row stay zero (a mel slot may reach bin 515)
    if (tid < 16 * 7) P[(tid / 7) * kProw + 513 + tid % 7] = 0.0f;
    // a wait the compiler sees for the loads above (features_v2.hip: loop-invariant loads carried as "pending")
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) asm volatile("" : "+v"(a1h[t][rb]), "+v"(a1l[t][rb]));
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) asm volatile("" : "+v"(a2h[rb]), "+v"(a2l[rb]));
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) asm volatile("" : "+v"(a2sh[rb]), "+v"(a2sl[rb]));

    // this wave's share of the mel plan
    int slot_end_w[NSLOT];
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) slot_end_w[q] = p.plan.slot_end[w][q];
    const int n_slots_w = p.plan.wave_nslots[w];
    const int n_mels = p.plan.n_mels;
    const bool vec_ok = (n_mels % 4 == 0) && (reinterpret_cast<uintptr_t>(p.out) % 16 == 0);
    const float* pbase = P + f * kProw + 4 * g;
    const unsigned row_off = (unsigned)f * (unsigned)n_mels + 4u * (unsigned)g;

    // lane constants of the tile update: sample pair p = tid + 512 r  ->  n2 = p & 31, q = p >> 5
    const int tn2 = tid & 31, tq = tid >> 5;                       // q of r = 0; r = 1: tq + 16; r = 2 (fresh): tq + 32

    const auto sample_pos = [&](int quad, int j) { return 32 * (4 * quad + j) + tn2; };
    const auto load_raw = [&](int64_t frame) -> raw_t {
        return *reinterpret_cast<const raw_t*>(static_cast<const char*>(p.pcm) + frame * S::kBpf);
    };

    bool fresh = true;
    UnitDesc ud_next = load_desc(p.units, u);
    float c_cur = 1.0f;            // 16 / s of the tile in LDS
    unsigned tail_prev = 0;        // bits of max |x| over the tile's last 768 samples
    int hb = 0;                    // halo buffer that holds those samples as f32
    raw_t pf[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) pf[r][j] = S::zero();

    // the new samples of tile (d.unit + k): tile positions [768, 4864) = PCM frames d.pcm0 + 4096 k + position
    const auto issue_prefetch = [&](const UnitDesc& d, int k) {
        const int64_t f0 = d.pcm0 + 4096 * k;
        const int n_left = d.n_avail - 4096 * k;
        if (n_left >= kTile) {                           // wave-uniform: the piece lies inside the clip
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) pf[r][j] = load_raw(f0 + sample_pos(6 + tq + 16 * r, j));
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int s = sample_pos(6 + tq + 16 * r, j);
                    pf[r][j] = s < n_left ? load_raw(f0 + s) : S::zero();
                }
        }
    };

    while (u < u_end) {
        const UnitDesc ud = ud_next;
        ud_next = load_desc(p.units, u + 1);             // (the table ends with one spare record)
        const int sel = SEP ? ud.sel : -1;
        const bool has_next = (u + 1 < u_end) && (ud.flags & 1);

        if (fresh) {
            // ---- a clip starts: the whole tile from global memory ----
            __syncthreads();                             // nobody still reads T, P, the halo or wmax
            float m[3][4];
            float mx = 0.0f, mt = 0.0f;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int quad = tq + 16 * r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int s = sample_pos(quad, j);
                    m[r][j] = (quad < kQuads && s < ud.n_avail) ? S::mono(load_raw(ud.pcm0 + s), sel) : 0.0f;
                    mx = fmaxf(mx, fabsf(m[r][j]));
                    if (r == 2) mt = fmaxf(mt, fabsf(m[r][j]));            // quads 32..37: the last 768 samples
                }
            }
            const unsigned wm = wave_max_bits(mx), wt = wave_max_bits(mt);
            if (lane == 0) { wmax[2 * w] = wm; wmax[2 * w + 1] = wt; }
            __syncthreads();
            unsigned mall = 0, mtl = 0;
#pragma unroll
            for (int ww = 0; ww < kW; ++ww) {
                mall = wmax[2 * ww] > mall ? wmax[2 * ww] : mall;
                mtl = wmax[2 * ww + 1] > mtl ? wmax[2 * ww + 1] : mtl;
            }
            const Scale sc = make_scale(mall);
            c_cur = sc.c;
            tail_prev = mtl;
            hb = 0;
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
            if (has_next) issue_prefetch(ud, 1);
            __syncthreads();                             // T ready
        }

        // ---- stage 1: this wave's four n2 ----
        f32x4 y[4][2];
        {
            const int bro = 16 * (f + g);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int n2 = 4 * w + t;
                const f16x8 bh = lds_frag(Thi + n2 * kTRow + bro);
                const f16x8 bl = lds_frag(Tlo + n2 * kTRow + bro);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 p1 = mfma16(a1h[t][rb], bh, z);
                    f32x4 p2 = mfma16(a1h[t][rb], bl, z);
                    p2 = mfma16(a1l[t][rb], bh, p2);
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[t][rb][i] = __builtin_fmaf(p2[i], 1.0f / 2048.0f, p1[i]);
                }
            }
            // -> f16 pairs -> Y[k1 = 4g + i][f][row block rb: Yr | Yi][hi | lo'][n2 = 4w .. 4w+3]
            unsigned char* yw = Ybase + (4 * g) * kYSlot + f * kYRow + 8 * w;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned h[2], l[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float a = y[2 * q][rb][i], b = y[2 * q + 1][rb][i];
                        const f16x2 hv = __builtin_convertvector(f32x2{a, b}, f16x2);
                        const float ra = __builtin_fmaf((float)hv[0], -2048.0f, a * 2048.0f);
                        const float rbb = __builtin_fmaf((float)hv[1], -2048.0f, b * 2048.0f);
                        const f16x2 lv = __builtin_convertvector(f32x2{ra, rbb}, f16x2);
                        h[q] = __builtin_bit_cast(unsigned, hv);
                        l[q] = __builtin_bit_cast(unsigned, lv);
                    }
                    *reinterpret_cast<uint2*>(yw + i * kYSlot + rb * 64) = uint2{h[0], h[1]};
                    *reinterpret_cast<uint2*>(yw + i * kYSlot + rb * 64 + 128) = uint2{l[0], l[1]};
                }
        }

        // ---- the prefetched piece: mono samples and their maxima, published before barrier 1 ----
        float mn[2][4];
        if (has_next) {
            float mx = 0.0f, mt = 0.0f;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mn[r][j] = S::mono(pf[r][j], sel);
                    mx = fmaxf(mx, fabsf(mn[r][j]));
                    if (r == 1) mt = fmaxf(mt, fabsf(mn[r][j]));
                }
            const unsigned wm = wave_max_bits(mx);
            // quads 32..37 (the next tile's last 768 samples) are r = 1 of threads 320..511 = waves 5..7
            const unsigned wt = w >= 5 ? wave_max_bits(mt) : 0u;
            if (lane == 0) { wmax[2 * w] = wm; wmax[2 * w + 1] = wt; }
        }
        __syncthreads();                                 // B1: Y ready, maxima published, T free

        float c_next = c_cur;
        unsigned tail_next = tail_prev;
        if (has_next) {
            // ---- tile update: the next unit's tile = the last 768 samples (f32 copy in the halo buffer) + the new piece ----
            unsigned mall = tail_prev, mtl = 0;
#pragma unroll
            for (int ww = 0; ww < kW; ++ww) {
                mall = wmax[2 * ww] > mall ? wmax[2 * ww] : mall;
                mtl = wmax[2 * ww + 1] > mtl ? wmax[2 * ww + 1] : mtl;
            }
            const Scale sc = make_scale(mall);
            c_next = sc.c;
            tail_next = mtl;
            if (tid < 192) {                             // quads 0..5 <- halo
                const float4 hv = *reinterpret_cast<const float4*>(Halo + hb * (32 * kHaloRow) + tn2 * kHaloRow + tq * 16);
                const float hm[4] = {hv.x, hv.y, hv.z, hv.w};
                uint2 hi, lo;
                split4(hm, sc, hi, lo);
                *reinterpret_cast<uint2*>(Thi + tn2 * kTRow + tq * 8) = hi;
                *reinterpret_cast<uint2*>(Tlo + tn2 * kTRow + tq * 8) = lo;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int quad = 6 + tq + 16 * r;
                uint2 hi, lo;
                split4(mn[r], sc, hi, lo);
                *reinterpret_cast<uint2*>(Thi + tn2 * kTRow + quad * 8) = hi;
                *reinterpret_cast<uint2*>(Tlo + tn2 * kTRow + quad * 8) = lo;
                if (r == 1 && w >= 5)
                    *reinterpret_cast<float4*>(Halo + (hb ^ 1) * (32 * kHaloRow) + tn2 * kHaloRow + (quad - 32) * 16) =
                        float4{mn[r][0], mn[r][1], mn[r][2], mn[r][3]};
            }
            // the piece after that, one unit ahead
            if ((u + 2 < u_end) && (ud_next.flags & 1)) issue_prefetch(ud, 2);
        }

        // ---- stage 2: two k1 per wave -> power rows ----
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int slot = (2 * w + 1 + it) & 15;      // k1; 0: the special item (k1 = 0 and k1 = 16)
            const unsigned char* yb = Ybase + slot * kYSlot + f * kYRow + 16 * g;
            const f16x8 yrh = lds_frag(yb), yih = lds_frag(yb + 64), yrl = lds_frag(yb + 128), yil = lds_frag(yb + 192);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            // row 4g + i of row block rb is k2 = 8 rb + 4 (g >> 1) + (g & 1) + 2 (i >> 1), cos (i even) or -sin (i odd):
            // the two lane groups of a 32-lane store half differ in the parity of k2, i.e. by 32 banks
            const int k2g = 4 * (g >> 1) + (g & 1);
            float* prow = P + f * kProw;
            if (slot != 0) {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    const f32x4 a0p1 = mfma16(a2h[rb], yrh, z);
                    const f32x4 a1p1 = mfma16(a2h[rb], yih, z);
                    f32x4 a0p2 = mfma16(a2h[rb], yrl, z);
                    f32x4 a1p2 = mfma16(a2h[rb], yil, z);
                    a0p2 = mfma16(a2l[rb], yrh, a0p2);
                    a1p2 = mfma16(a2l[rb], yih, a1p2);
                    float pw[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float re = __builtin_fmaf(a0p2[2 * h] - a1p2[2 * h + 1], 1.0f / 2048.0f, a0p1[2 * h] - a1p1[2 * h + 1]);
                        const float im = __builtin_fmaf(a1p2[2 * h] + a0p2[2 * h + 1], 1.0f / 2048.0f, a1p1[2 * h] + a0p1[2 * h + 1]);
                        pw[h] = __builtin_fmaf(re, re, im * im);
                    }
                    const int k2 = 8 * rb + k2g;         // and k2 + 2
                    if (rb < 2) {
                        prow[slot + 32 * k2] = pw[0];
                        prow[slot + 32 * k2 + 64] = pw[1];
                    } else {                             // bin 1024 - k
                        prow[1024 - slot - 32 * k2] = pw[0];
                        prow[1024 - slot - 32 * k2 - 64] = pw[1];
                    }
                }
            } else {
                // k1 = 0: X[32 k2] = A2 Y0 (real input), k2 = 0..16
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) {
                    const f32x4 p1 = mfma16(a2h[rb], yrh, z);
                    f32x4 p2 = mfma16(a2h[rb], yrl, z);
                    p2 = mfma16(a2l[rb], yrh, p2);
                    float pw[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float re = __builtin_fmaf(p2[2 * h], 1.0f / 2048.0f, p1[2 * h]);
                        const float im = __builtin_fmaf(p2[2 * h + 1], 1.0f / 2048.0f, p1[2 * h + 1]);
                        pw[h] = __builtin_fmaf(re, re, im * im);
                    }
                    const int k2 = 8 * rb + k2g;
                    if (rb < 2) {
                        prow[32 * k2] = pw[0];
                        prow[32 * k2 + 64] = pw[1];
                    } else if (g == 0) {
                        prow[512] = pw[0];               // k2 = 16: bin 512
                    }
                }
                // k1 = 16: X[16 + 32 k2] = A2s Y16, k2 = 0..15
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const f32x4 p1 = mfma16(a2sh[rb], yih, z);
                    f32x4 p2 = mfma16(a2sh[rb], yil, z);
                    p2 = mfma16(a2sl[rb], yih, p2);
                    float pw[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float re = __builtin_fmaf(p2[2 * h], 1.0f / 2048.0f, p1[2 * h]);
                        const float im = __builtin_fmaf(p2[2 * h + 1], 1.0f / 2048.0f, p1[2 * h + 1]);
                        pw[h] = __builtin_fmaf(re, re, im * im);
                    }
                    const int k2 = 8 * rb + k2g;
                    prow[16 + 32 * k2] = pw[0];
                    prow[16 + 32 * k2 + 64] = pw[1];
                }
            }
        }
        __syncthreads();                                 // B2: power rows ready; next tile ready; Y free

        // ---- mel: banded W (A, registers) x P^T (B, bins x 16 frames) on v_mfma_f32_16x16x4_f32 (features_v2.hip) ----
        {
            float wf[NSLOT * kSlotSteps], bop[NSLOT * kSlotSteps];
            const float4* wbase = reinterpret_cast<const float4*>(p.wfrag) + (size_t)w * kMaxSlotsW * 64;
            const int* lm = lmeta + w * kMaxSlotsW;
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) {
                const float4 v = wbase[lane + q * 64];
                wf[4 * q] = v.x; wf[4 * q + 1] = v.y; wf[4 * q + 2] = v.z; wf[4 * q + 3] = v.w;
                const float4 bv = *reinterpret_cast<const float4*>(pbase + lm[q]);
                bop[4 * q] = bv.x; bop[4 * q + 1] = bv.y; bop[4 * q + 2] = bv.z; bop[4 * q + 3] = bv.w;
            }
            float* obase = p.out + ud.out0 * n_mels;     // wave-uniform
            const bool fvalid = f < ud.n_rows;
            const int i0 = 4 * g;
            const float c2 = c_cur * c_cur;              // in range: c in [2^-122, 2^118] only for absurd inputs
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) {
                if (q < n_slots_w) {
#pragma unroll
                    for (int st = q * kSlotSteps; st < (q + 1) * kSlotSteps; st += 2) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[st], bop[st], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[st + 1], bop[st + 1], acc1, 0, 0, 0);
                    }
                    const int e = slot_end_w[q];         // wave-uniform (kernarg)
                    if (e != 0) {
                        f32x4 r = acc0 + acc1;
#pragma unroll
                        for (int i = 0; i < 4; ++i) r[i] = (r[i] * c_cur) * c_cur;      // the tile's scale leaves here
                        (void)c2;
                        if (p.log_mel) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) r[i] = __log10f(fmaxf(r[i], 1e-10f));
                        }
                        acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
                        acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
                        const int mm = e & 255, cnt = (e >> 8) & 31;
                        float* o = obase + mm + row_off;
                        if (vec_ok && ((cnt | mm) & 3) == 0) {
                            if (fvalid && i0 < cnt) *reinterpret_cast<float4*>(o) = float4{r[0], r[1], r[2], r[3]};
                        } else if (fvalid) {
                            if (i0 < cnt) o[0] = r[0];
                            if (i0 + 1 < cnt) o[1] = r[1];
                            if (i0 + 2 < cnt) o[2] = r[2];
                            if (i0 + 3 < cnt) o[3] = r[3];
                        }
                    }
                }
            }
        }

        c_cur = c_next;
        tail_prev = tail_next;
        if (has_next) hb ^= 1;
        fresh = !has_next;
        ++u;
    }
}

template <int CH, bool I16, bool SEP, int NSLOT>
static af_status launch_k(const Params& p, int grid, hipStream_t stream)
{
    auto kern = af_features_v3_kernel<CH, I16, SEP, NSLOT>;
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
    for (int w = 0; w < kW; ++w) most = p.plan.wave_nslots[w] > most ? p.plan.wave_nslots[w] : most;
    if (most > 8) return fail(AF_ERR_UNSUPPORTED, "features v3: %d mel slots on one wave (limit 8)", most);
    if constexpr (CH == 2) {
        if (separate) return launch_k<CH, I16, true, 8>(p, grid, stream);
    }
    if (most <= 6) return launch_k<CH, I16, false, 6>(p, grid, stream);
    return launch_k<CH, I16, false, 8>(p, grid, stream);
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
    // twiddle W1024^(16 n2) (it sits in a2s), rows 1..15: Im k1 = 1..15.  Scaled by 1/16 so that |Y'| <= 2^14 for a
    // tile whose maximum is below 2^14 (sum over n1 of the Hann window = 16).
    for (int n2 = 0; n2 < 32; ++n2)
        for (int rb = 0; rb < 2; ++rb) {
            uint4* dst = tabs + ((n2 * 2 + rb) * 2) * 64;
            put_frag(dst, dst + 64, [&](int m, int n1) {
                const int n = 32 * n1 + n2;
                const double wv = win[n] / 16.0;
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
