// features_pair_kernel.hip -- the fused  PCM -> to_mono -> Hann STFT(1024) -> |X|^2 -> mel
// kernel in its high-occupancy form (hop 256): ONE packed complex FFT (two real frames) per
// wave64 instead of two, so a lane keeps 16 complex points (32 VGPRs) instead of 32.  With
// <= 128 VGPRs and 8.5 KB of LDS per frame pair, two 8-wave workgroups = 16 waves share a CU
// (4 per SIMD), which is what keeps the fp32 VALU fed (the 4-wave kernel of features_kernel.hip
// issues from 2 waves per SIMD and leaves the pipe ~65 % idle).  Same arithmetic definition
// (docs/OPS.md), same LDS signal tile with register prefetch, same banded-MFMA mel phase.
//
// FFT: 1024 = 32 (n1) x 32 (n2).  A radix-32 is shared by a lane PAIR (lanes 2x and 2x+1,
// h = lane & 1): one cross-lane radix-2 step (DPP quad_perm, no LDS) followed by a radix-16
// in registers:  lane h owns n1 = 16h + a;  t[a] = other + (1-2h) * own;  h = 1 multiplies by
// W32^a;  DFT16 -> Y[k1 = 2k' + h].  Twiddle W1024^{n2 k1}, transpose through the wave's LDS
// buffer, same pair radix-32 over n2 -> Z[k1 + 32 (2k'' + h2)].
#include "features_dev.hpp"

namespace af {
namespace feat {

__device__ __forceinline__ constexpr int brev4(int x)
{
    return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3);
}

// cos / sin (2 pi j / 16), j = 0..7
__device__ static constexpr float kC16[8] = {1.0f, 0.923879504f, 0.707106769f, 0.382683426f,
                                             0.0f, -0.382683426f, -0.707106769f, -0.923879504f};
__device__ static constexpr float kS16[8] = {0.0f, 0.382683426f, 0.707106769f, 0.923879504f,
                                             1.0f, 0.923879504f, 0.707106769f, 0.382683426f};

template <int SPAN>
__device__ __forceinline__ void dif16_stage(float (&re)[16], float (&im)[16])
{
#pragma unroll
    for (int g = 0; g < 16; g += 2 * SPAN) {
#pragma unroll
        for (int j = 0; j < SPAN; ++j) {
            const int a = g + j, b = a + SPAN;
            const int t = j * (8 / SPAN);         // twiddle e^{-2 pi i t / 16}
            const float ar = re[a], ai = im[a], br = re[b], bi = im[b];
            re[a] = ar + br;
            im[a] = ai + bi;
            const float dr = ar - br, di = ai - bi;
            if (t == 0) {
                re[b] = dr;
                im[b] = di;
            } else if (t == 4) {                  // * (-i)
                re[b] = di;
                im[b] = -dr;
            } else if (t == 2) {                  // * (1 - i)/sqrt2
                re[b] = (dr + di) * 0.707106769f;
                im[b] = (di - dr) * 0.707106769f;
            } else if (t == 6) {                  // * (-1 - i)/sqrt2
                re[b] = (di - dr) * 0.707106769f;
                im[b] = (-dr - di) * 0.707106769f;
            } else {
                const float c = kC16[t], s = kS16[t];
                re[b] = dr * c + di * s;
                im[b] = di * c - dr * s;
            }
        }
    }
}

// 16-point DFT in registers; output X[k] lands in element brev4(k).
__device__ __forceinline__ void fft16(float (&re)[16], float (&im)[16])
{
    dif16_stage<8>(re, im);
    dif16_stage<4>(re, im);
    dif16_stage<2>(re, im);
    dif16_stage<1>(re, im);
}

// value held by the other lane of the pair (lane ^ 1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float pair_other(float v)
{
    const int i = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false);
    return __int_as_float(i);
}

// One radix-32 over the 32 points held by a lane pair (16 each, h = lane & 1 owns indices
// 16h + a): cross-lane DIF butterfly, twiddle on the odd branch, radix-16 in registers.
// Result: this lane's element brev4(k') is output 2k' + h.
__device__ __forceinline__ void pair_fft32(float (&re)[16], float (&im)[16], float sigma,
                                           const float* __restrict__ xtw_row)
{
    // t = other + sigma * own  (h = 0: x[a] + x[a+16];  h = 1: x[a] - x[a+16])
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const float orr = pair_other(re[a]), oi = pair_other(im[a]);
        re[a] = re[a] * sigma + orr;
        im[a] = im[a] * sigma + oi;
    }
    // h = 1: * W32^a; h = 0: * 1  (row h of a 2-row LDS table, two a per ds_read_b128)
    const float4* tr = reinterpret_cast<const float4*>(xtw_row);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 t = tr[j];
        if (j > 0) {
            const float xr = re[2 * j], xi = im[2 * j];
            re[2 * j] = xr * t.x + xi * t.y;
            im[2 * j] = xi * t.x - xr * t.y;
        }
        const float xr = re[2 * j + 1], xi = im[2 * j + 1];
        re[2 * j + 1] = xr * t.z + xi * t.w;
        im[2 * j + 1] = xi * t.z - xr * t.w;
    }
    fft16(re, im);
}

#define AF_STAMP_P(i)                                                       \
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

template <int CH, bool I16, bool STAMPS>
__global__ __launch_bounds__(kPairThreads, 4) void af_features_pair_kernel(const Params p)
{
    using St = Stage<CH, I16>;
    constexpr int HOP = 256;
    constexpr int MS = St::MS;
    constexpr int PFU = 16 * HOP / MS / kPairThreads;          // 16-byte prefetch units per thread
    static_assert(PFU >= 1 && (16 * HOP / MS) % kPairThreads == 0, "staging geometry");
    constexpr int kTileLen = 15 * HOP + kNfft;                 // 4864
    constexpr int kTail = kNfft - HOP;                         // 768

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                                        // kTileLen
    float* wbufs = tile + kTileLen;                            // 8 x kPairWbuf
    float* wtab = wbufs + kPairWaves * kPairWbuf;              // [64][kPairWinRow]
    float* ttab = wtab + 64 * kPairWinRow;                     // [64][kPairTwRow]
    float* xtab = ttab + 64 * kPairTwRow;                      // [2][kPairTwRow]
    int* lmeta = reinterpret_cast<int*>(xtab + 2 * kPairTwRow);  // [8][kPairSteps]
    float* partial = reinterpret_cast<float*>(lmeta + kPairWaves * kPairSteps);  // [n_groups][64][4]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane & 1, x = lane >> 1;                     // pair member, n2 resp. k1
    const float sigma = h ? -1.0f : 1.0f;
    const bool aligned = p.aligned16 != 0;

    const int64_t G = gridDim.x, b = blockIdx.x;
    int64_t u = p.total_units * b / G;
    const int64_t u_end = p.total_units * (b + 1) / G;
    if (u >= u_end) return;

    for (int i = tid; i < 64 * kPairWinRow / 4; i += kPairThreads)
        reinterpret_cast<float4*>(wtab)[i] = reinterpret_cast<const float4*>(p.pair_win)[i];
    for (int i = tid; i < 64 * kPairTwRow / 4; i += kPairThreads)
        reinterpret_cast<float4*>(ttab)[i] = reinterpret_cast<const float4*>(p.pair_tw)[i];
    for (int i = tid; i < 2 * kPairTwRow / 4; i += kPairThreads)
        reinterpret_cast<float4*>(xtab)[i] = reinterpret_cast<const float4*>(p.pair_xtw)[i];
    for (int i = tid; i < kPairWaves * kPairSteps; i += kPairThreads) lmeta[i] = p.pair_meta[i];

    // mel phase: every wave multiplies half of the K range of "its" filter groups
    // (wave w and w + 4 share the groups of plan wave w & 3)
    const bool lower = w < 4;
    const float* __restrict__ wfrag_w = p.pair_wfrag + (((size_t)w * kPairSlots) * 64 + lane) * 4;
    const int n_slots_w = p.pplan.nslots[w];
    int slot_end_w[kPairSlots];
#pragma unroll
    for (int q = 0; q < kPairSlots; ++q) slot_end_w[q] = p.pplan.slot_end[w][q];
    const int n_mels = p.plan.n_mels;
    const bool vec_ok = (n_mels % 4 == 0) && (reinterpret_cast<uintptr_t>(p.mel) % 16 == 0);

    int c;
    {
        int lo = 0, hi = p.n_clips;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (p.unit_off[mid] <= u) lo = mid; else hi = mid;
        }
        c = lo;
    }
    // experimental stagger (AF_FEATURES_STAGGER): delay a subset of workgroups by about half a
    // unit so that co-resident workgroups alternate VALU-heavy and MFMA/barrier phases
    if (p.stagger_mode == 1 && (blockIdx.x & 1)) for (int i = 0; i < p.stagger_len; ++i) __builtin_amdgcn_s_sleep(64);
    if (p.stagger_mode == 2 && blockIdx.x >= gridDim.x / 2) for (int i = 0; i < p.stagger_len; ++i) __builtin_amdgcn_s_sleep(64);
    bool fresh = true;
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = STAMPS ? __builtin_amdgcn_s_memtime() : 0;

    float* wbuf = wbufs + w * kPairWbuf;

    while (u < u_end) {
        while (p.unit_off[c + 1] <= u) ++c;
        const int64_t c_pcm0 = p.pcm_off[c];
        const int64_t c_n = p.pcm_len[c];
        const int64_t c_out0 = p.out_off[c];
        const int64_t c_frames = p.out_off[c + 1] - c_out0;
        const int64_t c_u1 = p.unit_off[c + 1];
        const int64_t F0 = (u - p.unit_off[c]) * kFramesPerIter;

        if (fresh) {
            __syncthreads();                      // nobody still reads the old tile
            const int64_t pos0 = F0 * HOP;
            for (int v = tid; v * MS < kTileLen; v += kPairThreads) {
                float m[MS];
                St::load(p.pcm, c_pcm0, pos0 + (int64_t)v * MS, c_n, aligned, m);
                St::store(tile + v * MS, m);
            }
        }
        __syncthreads();                          // B1: tile ready (and last unit's power rows consumed)
        AF_STAMP_P(0);

        // prefetch the 16*HOP new samples of the next unit, raw, into registers
        const bool has_next = (u + 1 < u_end) && (u + 1 < c_u1);
        const int64_t next_pos0 = (F0 + kFramesPerIter) * HOP + kTail;
        uint4 pf[PFU];
        float tailr[2];
        if (has_next) {
            if (aligned && next_pos0 + 16 * HOP <= c_n) {
                const char* src = static_cast<const char*>(p.pcm) + (c_pcm0 + next_pos0) * St::kBytesPerFrame;
#pragma unroll
                for (int i = 0; i < PFU; ++i)
                    pf[i] = *reinterpret_cast<const uint4*>(src + (unsigned)(tid + i * kPairThreads) * 16u);
            } else {
#pragma unroll
                for (int i = 0; i < PFU; ++i)
                    pf[i] = St::load_raw(p.pcm, c_pcm0, next_pos0 + (int64_t)(tid + i * kPairThreads) * MS,
                                         c_n, aligned);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = tid + i * kPairThreads;
                tailr[i] = (j < kTail) ? tile[16 * HOP + j] : 0.0f;
            }
        }
        AF_STAMP_P(1);

        float wf[kPairSteps];
        {
            float ur[16], ui[16];
            {
                // ---- stage A: lane (n2 = x, h) owns n1 = 16h + a
                float re[16], im[16];
                {
                    float win[16];
                    const float4* wrow = reinterpret_cast<const float4*>(wtab + lane * kPairWinRow);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 v = wrow[j];
                        win[4 * j] = v.x; win[4 * j + 1] = v.y; win[4 * j + 2] = v.z; win[4 * j + 3] = v.w;
                    }
                    // frame a = 2w, frame b = 2w + 1 = frame a shifted by HOP = 8 steps of 32
                    const float* src = tile + 2 * w * HOP + 512 * h + x;
                    float s[24];
#pragma unroll
                    for (int j = 0; j < 24; ++j) s[j] = src[32 * j];
#pragma unroll
                    for (int a = 0; a < 16; ++a) {
                        re[a] = s[a] * win[a];
                        im[a] = s[a + 8] * win[a];
                    }
                }
                pair_fft32(re, im, sigma, xtab + h * kPairTwRow);
                // inter-stage twiddle e^{-2 pi i n2 (2k'+h) / 1024}
                {
                    const float4* trow = reinterpret_cast<const float4*>(ttab + lane * kPairTwRow);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float4 t = trow[j];
                        {
                            const int r = brev4(2 * j);
                            const float xr = re[r], xi = im[r];
                            re[r] = xr * t.x + xi * t.y;
                            im[r] = xi * t.x - xr * t.y;
                        }
                        {
                            const int r = brev4(2 * j + 1);
                            const float xr = re[r], xi = im[r];
                            re[r] = xr * t.z + xi * t.w;
                            im[r] = xi * t.z - xr * t.w;
                        }
                    }
                }
                // transpose Y[k1 = 2k'+h][n2 = x] -> lane (k1, h2) reads Y[k1][16 h2 + a]
#pragma unroll
                for (int k = 0; k < 16; ++k) wbuf[(2 * k + h) * kXStride + x] = re[brev4(k)];
                wave_sync();
#pragma unroll
                for (int a = 0; a < 16; ++a) ur[a] = wbuf[x * kXStride + 16 * h + a];
                wave_sync();
#pragma unroll
                for (int k = 0; k < 16; ++k) wbuf[(2 * k + h) * kXStride + x] = im[brev4(k)];
                wave_sync();
#pragma unroll
                for (int a = 0; a < 16; ++a) ui[a] = wbuf[x * kXStride + 16 * h + a];
                wave_sync();
            }

            // every wave fetches its filterbank fragments here (L2, 6 float4 per lane):
            // register pressure is lowest between the two radix-32 stages and a global load
            // needs microseconds to return while the chip streams PCM
            {
                int opaque_zero;
                asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
                const float4* wp = reinterpret_cast<const float4*>(wfrag_w) + opaque_zero;
#pragma unroll
                for (int q = 0; q < kPairSlots; ++q) {
                    const float4 v = wp[q * 64];
                    wf[4 * q] = v.x; wf[4 * q + 1] = v.y; wf[4 * q + 2] = v.z; wf[4 * q + 3] = v.w;
                }
            }

            // ---- stage B: lane (k1 = x, h2 = h); Z[k1 + 32 (2k''+h2)] at element brev4(k'')
            pair_fft32(ur, ui, sigma, xtab + h * kPairTwRow);

            // split the packed spectrum (docs in features_kernel.hip).  The partner Z[1024-k]
            // of element k'' sits in lane (32-k1, 1-h2), element 15-k''; the two k1 = 0 lanes
            // pair with themselves (lane (0,0): element 16-k'').  Publish elements {0, 8..15}
            // -> slots {0, 1..8} of a 9-dword row per lane.
            const int plane = 2 * ((32 - x) & 31) + ((x == 0) ? h : 1 - h);
            const int d = (lane == 0) ? 1 : 0;                 // lane (k1 = 0, h2 = 0)
            float* mine = wbuf + lane * 9;
            const float* theirs = wbuf + plane * 9 + d;
            const int first = d ? 0 : 8;
            float pr[9], pi[9];
            mine[0] = ur[brev4(0)];
#pragma unroll
            for (int k = 8; k < 16; ++k) mine[k - 7] = ur[brev4(k)];
            wave_sync();
            pr[0] = wbuf[plane * 9 + first];
#pragma unroll
            for (int q = 1; q <= 8; ++q) pr[q] = theirs[8 - q];
            wave_sync();
            mine[0] = ui[brev4(0)];
#pragma unroll
            for (int k = 8; k < 16; ++k) mine[k - 7] = ui[brev4(k)];
            wave_sync();
            pi[0] = wbuf[plane * 9 + first];
#pragma unroll
            for (int q = 1; q <= 8; ++q) pi[q] = theirs[8 - q];
            wave_sync();

            float* prow_a = wbuf + x + 32 * h;                 // bin = k1 + 32 h2 + 64 q
            float* prow_b = prow_a + kProw;
#pragma unroll
            for (int q = 0; q <= 8; ++q) {
                const float zr = ur[brev4(q)], zi = ui[brev4(q)];
                const float a1 = zr + pr[q], a2 = zi - pi[q];
                const float b1 = zi + pi[q], b2 = zr - pr[q];
                const float pa = a1 * a1 + a2 * a2;
                const float pb = b1 * b1 + b2 * b2;
                if (q < 8) {
                    prow_a[64 * q] = pa;
                    prow_b[64 * q] = pb;
                } else if (lane == 0) {                        // Nyquist bin 512
                    prow_a[512] = pa;
                    prow_b[512] = pb;
                }
            }
        }
        AF_STAMP_P(2);

        __syncthreads();                          // B2: power rows of the 16 frames ready, tile reads done
        AF_STAMP_P(3);

        if (has_next) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = tid + i * kPairThreads;
                if (j < kTail) tile[j] = tailr[i];
            }
#pragma unroll
            for (int i = 0; i < PFU; ++i) {
                float m[MS];
                St::convert(pf[i], m);
                St::store(tile + kTail + (tid + i * kPairThreads) * MS, m);
            }
        }
        AF_STAMP_P(4);

        {
            // B operands of this wave's K-steps, one branch-free batch
            const int j = lane & 15, kq = lane >> 4;
            const float* pbase = wbufs + (j >> 1) * kPairWbuf + (j & 1) * kProw + kq;
            const int* lm = lmeta + w * kPairSteps;
            float bop[kPairSteps];
            {
                int meta[kPairSteps];
#pragma unroll
                for (int s = 0; s < kPairSteps; ++s) meta[s] = lm[s];
#pragma unroll
                for (int s = 0; s < kPairSteps; ++s) bop[s] = pbase[meta[s]];
            }
            AF_STAMP_P(5);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            f32x4 held[4];                        // lower waves: finished groups wait for their partner half
#pragma unroll
            for (int i = 0; i < 4; ++i) held[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < kPairSlots; ++q) {
                if (q < n_slots_w) {
#pragma unroll
                    for (int s = 4 * q; s < 4 * q + 4; s += 2) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], bop[s], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s + 1], bop[s + 1], acc1, 0, 0, 0);
                    }
                    const int e = slot_end_w[q];  // wave-uniform (kernarg)
                    if (e != 0) {
                        const f32x4 r = acc0 + acc1;
                        acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
                        acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (lower) {
                            const int ord = (e >> 28) & 3;
                            if (ord == 0) held[0] = r;
                            else if (ord == 1) held[1] = r;
                            else if (ord == 2) held[2] = r;
                            else held[3] = r;
                        } else {
                            const int gid = (e >> 24) & 15;
                            *reinterpret_cast<float4*>(partial + (gid * 64 + lane) * 4) =
                                float4{r[0], r[1], r[2], r[3]};
                        }
                    }
                }
            }
            __syncthreads();                      // B3: upper halves' partial tiles are in LDS
            if (lower) {
                float* orow = p.mel + (c_out0 + F0 + j) * n_mels;
                const bool fvalid = F0 + j < c_frames;
#pragma unroll
                for (int q = 0; q < kPairSlots; ++q) {
                    if (q < n_slots_w && slot_end_w[q] != 0) {
                        const int e = slot_end_w[q];
                        const int ord = (e >> 28) & 3;
                        f32x4 r = ord == 0 ? held[0] : ord == 1 ? held[1] : ord == 2 ? held[2] : held[3];
                        if ((e >> 30) & 1) {
                            const int gid = (e >> 24) & 15;
                            const float4 o = *reinterpret_cast<const float4*>(partial + (gid * 64 + lane) * 4);
                            r[0] += o.x; r[1] += o.y; r[2] += o.z; r[3] += o.w;
                        }
                        const int m0 = e & 255, cnt = (e >> 8) & 31;
                        const int i0 = 4 * kq;
                        float* o = orow + m0 + i0;
                        if (vec_ok && (cnt & 3) == 0) {       // wave-uniform choice
                            if (fvalid && i0 < cnt)
                                *reinterpret_cast<float4*>(o) = float4{r[0], r[1], r[2], r[3]};
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
        AF_STAMP_P(6);
        ++u;
        fresh = !has_next;
    }
    if (STAMPS && lane == 0 && w < 4)
        for (int i = 0; i < 8; ++i) atomicAdd(p.stamps + w * 8 + i, acc_t[i]);
}

size_t features_pair_smem_bytes()
{
    return sizeof(float) * ((size_t)(15 * 256 + kNfft) + kPairWaves * kPairWbuf + 64 * kPairWinRow +
                            64 * kPairTwRow + 2 * kPairTwRow + kPairWaves * kPairSteps +
                            kPairMaxGroups * 64 * 4);
}

template <int CH, bool I16, bool STAMPS>
static af_status launch_pk(const Params& p, int grid, hipStream_t stream)
{
    auto kern = af_features_pair_kernel<CH, I16, STAMPS>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        AF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kPairThreads), features_pair_smem_bytes(), stream, p);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

template <int CH, bool I16>
static af_status launch_p(const Params& p, int grid, hipStream_t stream)
{
    return p.stamps ? launch_pk<CH, I16, true>(p, grid, stream) : launch_pk<CH, I16, false>(p, grid, stream);
}

af_status launch_features_pair(const Params& p, int channels, bool i16, int grid, hipStream_t stream)
{
    if (grid <= 0) return AF_OK;
    if (channels == 2 && !i16) return launch_p<2, false>(p, grid, stream);
    if (channels == 1 && !i16) return launch_p<1, false>(p, grid, stream);
    if (channels == 2 && i16) return launch_p<2, true>(p, grid, stream);
    if (channels == 1 && i16) return launch_p<1, true>(p, grid, stream);
    return fail(AF_ERR_UNSUPPORTED, "features: channels must be 1 or 2 (got %d)", channels);
}

}  // namespace feat
}  // namespace af
