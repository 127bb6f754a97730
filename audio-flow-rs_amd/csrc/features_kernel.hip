// features_kernel.hip -- fused  interleaved PCM -> to_mono -> Hann STFT(1024)
// -> |X|^2 -> mel filterbank  for gfx950 (MI355X).
//
// Replaces, on the device, AudioFrame::to_mono (reference
// src-tauri/src/modules/audio/capture.rs:30-42) followed by the build-defined
// STFT and mel ops of docs/OPS.md (the reference has no FFT/mel code at all,
// SURVEY.md section 0).
//
// Shape of the computation (one 256-thread workgroup = 4 wave64):
//   * a workgroup walks a contiguous run of 16-frame "units" of a clip.  The
//     mono signal tile (15*hop + 1024 samples) lives in LDS; only the 16*hop
//     new samples per unit are fetched from HBM (16-byte loads of interleaved
//     PCM, downmixed in registers, prefetched under the FFT of the previous
//     unit), so every input sample crosses HBM once despite the 75 % overlap.
//   * FFT: two real frames (f, f+1) are packed as re/im of ONE 1024-point
//     complex FFT done as 32 x 32: each half-wave (32 lanes) owns a frame pair;
//     lane n2 does a radix-32 DFT over n1 fully in registers, twiddles, one
//     transpose through LDS, lane k1 does the second radix-32.  The split into
//     the two real spectra needs Z[k] and Z[1024-k]; the partner values go
//     through the same LDS buffer.  Power is written to LDS ([frame][514]).
//   * mel: P^T (bins x 16 frames) is the B operand and the banded filterbank
//     the A operand of v_mfma_f32_16x16x4_f32 (exact f32 FMA chain).  Only the
//     non-zero band of each 16-filter tile is multiplied (about 134 K-steps of 4 bins
//     instead of 645 dense); the K-steps are split evenly over the 4 waves,
//     partial 16x16 tiles meet in LDS and are written out coalesced.
#include "features.hpp"

namespace af {
namespace feat {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ constexpr int brev5(int x)
{
    return ((x & 1) << 4) | ((x & 2) << 2) | (x & 4) | ((x & 8) >> 2) | ((x & 16) >> 4);
}

// cos / sin (2 pi j / 32), j = 0..15
__device__ static constexpr float kC32[16] = {
    1.0f, 0.980785251f, 0.923879504f, 0.831469595f, 0.707106769f, 0.555570245f,
    0.382683426f, 0.195090324f, 0.0f, -0.195090324f, -0.382683426f, -0.555570245f,
    -0.707106769f, -0.831469595f, -0.923879504f, -0.980785251f};
__device__ static constexpr float kS32[16] = {
    0.0f, 0.195090324f, 0.382683426f, 0.555570245f, 0.707106769f, 0.831469595f,
    0.923879504f, 0.980785251f, 1.0f, 0.980785251f, 0.923879504f, 0.831469595f,
    0.707106769f, 0.555570245f, 0.382683426f, 0.195090324f};

// One decimation-in-frequency stage over 32 register-resident complex points.
// Everything is compile-time indexed after unrolling.
template <int SPAN>
__device__ __forceinline__ void dif_stage(float (&re)[32], float (&im)[32])
{
#pragma unroll
    for (int g = 0; g < 32; g += 2 * SPAN) {
#pragma unroll
        for (int j = 0; j < SPAN; ++j) {
            const int a = g + j, b = a + SPAN;
            const int t = j * (16 / SPAN);        // twiddle e^{-2 pi i t / 32}
            const float ar = re[a], ai = im[a], br = re[b], bi = im[b];
            re[a] = ar + br;
            im[a] = ai + bi;
            const float dr = ar - br, di = ai - bi;
            if (t == 0) {
                re[b] = dr;
                im[b] = di;
            } else if (t == 8) {                  // * (-i)
                re[b] = di;
                im[b] = -dr;
            } else if (t == 4) {                  // * (1 - i)/sqrt2
                re[b] = (dr + di) * 0.707106769f;
                im[b] = (di - dr) * 0.707106769f;
            } else if (t == 12) {                 // * (-1 - i)/sqrt2
                re[b] = (di - dr) * 0.707106769f;
                im[b] = (-dr - di) * 0.707106769f;
            } else {                              // (dr + i di)(c - i s)
                const float c = kC32[t], s = kS32[t];
                re[b] = dr * c + di * s;
                im[b] = di * c - dr * s;
            }
        }
    }
}

// 32-point DFT in registers; output X[k] lands in element brev5(k).
__device__ __forceinline__ void fft32(float (&re)[32], float (&im)[32])
{
    dif_stage<16>(re, im);
    dif_stage<8>(re, im);
    dif_stage<4>(re, im);
    dif_stage<2>(re, im);
    dif_stage<1>(re, im);
}

// LDS traffic inside one wave needs no s_barrier (a wave's DS operations are
// executed in order); this only stops the compiler from moving LDS accesses
// across the hand-off.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------
// PCM staging: one 16-byte unit of interleaved PCM -> MS mono samples.
// to_mono semantics (capture.rs:34-40): sequential f32 sum over the channels,
// then divided by the channel count (x0.5 is exact for stereo).
// i16 samples are scaled by 1/32768 (docs/OPS.md "PCM formats").
template <int CH, bool I16>
struct Stage {
    static constexpr int kBytesPerFrame = CH * (I16 ? 2 : 4);
    static constexpr int MS = 16 / kBytesPerFrame;   // mono samples per 16-B unit

    __device__ static __forceinline__ float sample(const void* pcm, int64_t elem)
    {
        if (I16) return (float)static_cast<const int16_t*>(pcm)[elem] * (1.0f / 32768.0f);
        return static_cast<const float*>(pcm)[elem];
    }

    // mono samples [pos, pos+MS) of a clip whose first frame is `base` and
    // which has `n` frames; zero beyond the end of the clip
    __device__ static __forceinline__ void load(const void* pcm, int64_t base, int64_t pos,
                                                int64_t n, bool aligned, float (&m)[MS])
    {
        if (aligned && pos + MS <= n) {
            const uint4 raw = *reinterpret_cast<const uint4*>(
                static_cast<const char*>(pcm) + (base + pos) * kBytesPerFrame);
            const uint32_t r[4] = {raw.x, raw.y, raw.z, raw.w};
            if constexpr (!I16) {
                float f[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(r[i]);
                if constexpr (CH == 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) m[i] = f[i];
                } else {
                    m[0] = (f[0] + f[1]) * 0.5f;
                    m[1] = (f[2] + f[3]) * 0.5f;
                }
            } else {
                float f[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f[2 * i] = (float)(int16_t)(r[i] & 0xffffu) * (1.0f / 32768.0f);
                    f[2 * i + 1] = (float)(int16_t)(r[i] >> 16) * (1.0f / 32768.0f);
                }
                if constexpr (CH == 1) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) m[i] = f[i];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) m[i] = (f[2 * i] + f[2 * i + 1]) * 0.5f;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < MS; ++i) {
                float v = 0.0f;
                if (pos + i < n) {
                    const int64_t e = (base + pos + i) * CH;
                    float s = sample(pcm, e);
                    if (CH == 2) s = (s + sample(pcm, e + 1)) * 0.5f;
                    v = s;
                }
                m[i] = v;
            }
        }
    }

    __device__ static __forceinline__ void store(float* dst, const float (&m)[MS])
    {
#pragma unroll
        for (int i = 0; i < MS; ++i) dst[i] = m[i];
    }
};

// ---------------------------------------------------------------------------
// FFT phase of one wave: 4 frames (two per half-wave) -> 4 power rows in LDS.
template <int HOP_T>
__device__ __forceinline__ void fft_phase(const float* __restrict__ tile, float* wbuf, int hop,
                                          int w, int lane, const float (&win)[32],
                                          const float (&twc)[32], const float (&tws)[32])
{
    const int half = lane >> 5, n2 = lane & 31;
    const int fa = 4 * w + 2 * half;              // first frame of this half's pair
    float re[32], im[32];
    {
        const float* src = tile + fa * hop + n2;
        if constexpr (HOP_T != 0 && HOP_T % 32 == 0) {
            // frame b = frame a shifted by HOP_T samples = SH registers
            constexpr int SH = HOP_T / 32;
            float s[32 + SH];
#pragma unroll
            for (int j = 0; j < 32 + SH; ++j) s[j] = src[32 * j];
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) {
                re[n1] = s[n1] * win[n1];
                im[n1] = s[n1 + SH] * win[n1];
            }
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) {
                re[n1] = src[32 * n1] * win[n1];
                im[n1] = src[hop + 32 * n1] * win[n1];
            }
        }
    }

    fft32(re, im);                                // over n1 -> Y[k1] at brev5(k1)

    // inter-stage twiddle e^{-2 pi i n2 k1 / 1024}
#pragma unroll
    for (int k1 = 1; k1 < 32; ++k1) {
        const int r = brev5(k1);
        const float xr = re[r], xi = im[r];
        re[r] = xr * twc[k1] + xi * tws[k1];
        im[r] = xi * twc[k1] - xr * tws[k1];
    }

    // transpose (k1, n2) through LDS, real parts then imaginary parts
    float* xb = wbuf + half * kHalfDw;
    float ur[32], ui[32];
#pragma unroll
    for (int k1 = 0; k1 < 32; ++k1) xb[k1 * kXStride + n2] = re[brev5(k1)];
    wave_sync();
#pragma unroll
    for (int n = 0; n < 32; ++n) ur[n] = xb[n2 * kXStride + n];
    wave_sync();
#pragma unroll
    for (int k1 = 0; k1 < 32; ++k1) xb[k1 * kXStride + n2] = im[brev5(k1)];
    wave_sync();
#pragma unroll
    for (int n = 0; n < 32; ++n) ui[n] = xb[n2 * kXStride + n];
    wave_sync();

    fft32(ur, ui);                                // over n2 -> Z[k1 + 32 k2] at brev5(k2)

    // Split the packed spectrum: A[k] = (Z[k] + conj Z[N-k])/2 (frame a),
    // B[k] = (Z[k] - conj Z[N-k])/(2i) (frame b); the 1/2 is folded into the
    // window table.  Lane k1 needs Z[N-k] = lane (32-k1)%32, k2' = 31-q
    // (lane 0: k2' = (32-q)%32).  Publish k2 in {0,16..31} -> slots {0,1..16}.
    const int k1 = n2;
    const int d = (k1 == 0) ? 1 : 0;
    float* mine = xb + k1 * kXStride;
    const float* theirs = xb + ((32 - k1) & 31) * kXStride + d;
    const int first = d ? 0 : 16;                 // slot of the q = 0 partner
    float pr[17], pi[17];

    mine[0] = ur[brev5(0)];
#pragma unroll
    for (int k2 = 16; k2 < 32; ++k2) mine[k2 - 15] = ur[brev5(k2)];
    wave_sync();
    pr[0] = xb[((32 - k1) & 31) * kXStride + first];
#pragma unroll
    for (int q = 1; q <= 16; ++q) pr[q] = theirs[16 - q];
    wave_sync();
    mine[0] = ui[brev5(0)];
#pragma unroll
    for (int k2 = 16; k2 < 32; ++k2) mine[k2 - 15] = ui[brev5(k2)];
    wave_sync();
    pi[0] = xb[((32 - k1) & 31) * kXStride + first];
#pragma unroll
    for (int q = 1; q <= 16; ++q) pi[q] = theirs[16 - q];
    wave_sync();

    float* prow_a = wbuf + (2 * half) * kProw + k1;
    float* prow_b = prow_a + kProw;
#pragma unroll
    for (int q = 0; q <= 16; ++q) {
        const float zr = ur[brev5(q)], zi = ui[brev5(q)];
        const float a1 = zr + pr[q], a2 = zi - pi[q];
        const float b1 = zi + pi[q], b2 = zr - pr[q];
        const float pa = a1 * a1 + a2 * a2;
        const float pb = b1 * b1 + b2 * b2;
        if (q < 16) {
            prow_a[32 * q] = pa;
            prow_b[32 * q] = pb;
        } else if (k1 == 0) {                     // Nyquist bin 512
            prow_a[512] = pa;
            prow_b[512] = pb;
        }
    }
}

// ---------------------------------------------------------------------------
// mel phase: this wave's share of the banded W (A) x P^T (B) MFMA steps.
__device__ __forceinline__ void mel_phase(const float* wbufs, float* partials,
                                          const Plan* __restrict__ plan,
                                          const float* __restrict__ wfrag, int w, int lane)
{
    const int f = lane & 15, kq = lane >> 4;
    const float* pbase = wbufs + (f >> 2) * kWbuf + (f & 3) * kProw + kq;
    const int nseg = plan->wave_nseg[w];
    for (int g = 0; g < nseg; ++g) {
        const int slot = plan->seg_slot[w][g];
        const int bin0 = plan->seg_bin0[w][g];
        const int step0 = plan->seg_step0[w][g];
        const int ns = plan->seg_nsteps[w][g];
        const float* pw = wfrag + (size_t)step0 * 64 + lane;
        const float* pp = pbase + bin0;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 2 <= ns; s += 2) {
            const float a0 = pw[s * 64], a1 = pw[s * 64 + 64];
            const float b0 = pp[4 * s], b1 = pp[4 * s + 4];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
        }
        if (s < ns)
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[s * 64], pp[4 * s], acc0, 0, 0, 0);
        acc0 += acc1;
        // D[i = mel row 4*kq + r][j = frame f]  ->  slot[j][i]
        float* ps = partials + slot * kSlotDw + f * 17 + 4 * kq;
        ps[0] = acc0[0];
        ps[1] = acc0[1];
        ps[2] = acc0[2];
        ps[3] = acc0[3];
    }
}

// ---------------------------------------------------------------------------
template <int HOP_T, int CH, bool I16>
__global__ __launch_bounds__(kThreads, 2) void af_features_kernel(const Params p)
{
    using St = Stage<CH, I16>;
    constexpr int MS = St::MS;
    // prefetch units per thread and iteration (compile-time only for HOP_T != 0)
    constexpr int PFU = HOP_T ? (16 * HOP_T / MS + kThreads - 1) / kThreads : 1;
    constexpr bool kPrefetch = HOP_T != 0 && PFU * MS <= 32;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int hop = HOP_T ? HOP_T : p.hop;
    const int tile_len = 15 * hop + kNfft;
    const int tile_alloc = (tile_len + 7) & ~7;
    const int tail_len = kNfft - hop;             // samples shared with the next unit
    float* tile = smem;
    float* wbufs = smem + tile_alloc;
    float* partials = wbufs + kWaves * kWbuf;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n2 = lane & 31;
    const bool aligned = p.aligned16 != 0;

    // work range of this workgroup, in 16-frame units
    const int64_t G = gridDim.x, b = blockIdx.x;
    int64_t u = p.total_units * b / G;
    const int64_t u_end = p.total_units * (b + 1) / G;
    if (u >= u_end) return;

    // per-lane constants: window column and inter-stage twiddles
    float win[32], twc[32], tws[32];
#pragma unroll
    for (int n1 = 0; n1 < 32; ++n1) win[n1] = p.wintab[32 * n1 + n2];
    twc[0] = 1.0f;
    tws[0] = 0.0f;
#pragma unroll
    for (int k1 = 1; k1 < 32; ++k1) {
        const float2 t = p.twtab[k1 * 32 + n2];
        twc[k1] = t.x;
        tws[k1] = t.y;
    }

    // clip containing unit u: largest c with unit_off[c] <= u
    int c;
    {
        int lo = 0, hi = p.n_clips;               // invariant: unit_off[lo] <= u < unit_off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (p.unit_off[mid] <= u) lo = mid; else hi = mid;
        }
        c = lo;
    }
    const int n_mels = p.plan->n_mels;
    bool fresh = true;

    while (u < u_end) {
        while (p.unit_off[c + 1] <= u) ++c;       // skip clips without frames
        const int64_t c_pcm0 = p.pcm_off[c];
        const int64_t c_n = p.pcm_off[c + 1] - c_pcm0;
        const int64_t c_out0 = p.out_off[c];
        const int64_t c_frames = p.out_off[c + 1] - c_out0;
        const int64_t c_u1 = p.unit_off[c + 1];
        const int64_t F0 = (u - p.unit_off[c]) * kFramesPerIter;

        if (fresh) {
            __syncthreads();                      // nobody still reads the old tile
            const int64_t pos0 = F0 * hop;
            for (int v = tid; v * MS < tile_alloc; v += kThreads) {
                float m[MS];
                St::load(p.pcm, c_pcm0, pos0 + (int64_t)v * MS, c_n, aligned, m);
                St::store(tile + v * MS, m);
            }
        }
        __syncthreads();                          // B1: tile ready

        const bool has_next = (u + 1 < u_end) && (u + 1 < c_u1);
        const int64_t next_pos0 = (F0 + kFramesPerIter) * hop + tail_len;  // first new sample
        float pf[PFU][MS];
        float tailr[4];
        if (has_next) {
            if constexpr (kPrefetch) {
#pragma unroll
                for (int i = 0; i < PFU; ++i) {
                    const int v = tid + i * kThreads;
                    if (v * MS < 16 * hop)
                        St::load(p.pcm, c_pcm0, next_pos0 + (int64_t)v * MS, c_n, aligned, pf[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = tid + i * kThreads;
                tailr[i] = (j < tail_len) ? tile[16 * hop + j] : 0.0f;
            }
        }

        fft_phase<HOP_T>(tile, wbufs + w * kWbuf, hop, w, lane, win, twc, tws);

        __syncthreads();                          // B2: power rows ready, tile reads done

        if (has_next) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = tid + i * kThreads;
                if (j < tail_len) tile[j] = tailr[i];
            }
            if constexpr (kPrefetch) {
#pragma unroll
                for (int i = 0; i < PFU; ++i) {
                    const int v = tid + i * kThreads;
                    if (v * MS < 16 * hop) St::store(tile + tail_len + v * MS, pf[i]);
                }
            } else {
                for (int v = tid; v * MS < 16 * hop; v += kThreads) {
                    float m[MS];
                    St::load(p.pcm, c_pcm0, next_pos0 + (int64_t)v * MS, c_n, aligned, m);
                    St::store(tile + tail_len + v * MS, m);
                }
            }
        }

        mel_phase(wbufs, partials, p.plan, p.wfrag, w, lane);

        __syncthreads();                          // B3: partial tiles ready

        for (int idx = tid; idx < kFramesPerIter * n_mels; idx += kThreads) {
            const int j = idx / n_mels;
            const int m = idx - j * n_mels;
            const int t = m >> 4, i = m & 15;
            const int s0 = p.plan->tile_slot0[t], sn = p.plan->tile_nslots[t];
            float sum = 0.0f;
            for (int q = 0; q < sn; ++q) sum += partials[(s0 + q) * kSlotDw + j * 17 + i];
            if (F0 + j < c_frames) p.mel[(c_out0 + F0 + j) * n_mels + m] = sum;
        }

        ++u;
        fresh = !has_next;
    }
}

size_t features_smem_bytes(int hop, int n_slots)
{
    const int tile_len = 15 * hop + kNfft;
    const int tile_alloc = (tile_len + 7) & ~7;
    return sizeof(float) * ((size_t)tile_alloc + kWaves * kWbuf + (size_t)n_slots * kSlotDw);
}

template <int HOP_T, int CH, bool I16>
static af_status launch_one(const Params& p, int grid, size_t smem, hipStream_t stream)
{
    auto kern = af_features_kernel<HOP_T, CH, I16>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        AF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), smem, stream, p);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

af_status launch_features(const Params& p, int channels, bool i16, int grid, size_t smem,
                          hipStream_t stream)
{
    if (grid <= 0) return AF_OK;
    const bool h256 = p.hop == 256;
    if (channels == 2 && !i16)
        return h256 ? launch_one<256, 2, false>(p, grid, smem, stream)
                    : launch_one<0, 2, false>(p, grid, smem, stream);
    if (channels == 1 && !i16)
        return h256 ? launch_one<256, 1, false>(p, grid, smem, stream)
                    : launch_one<0, 1, false>(p, grid, smem, stream);
    if (channels == 2 && i16)
        return h256 ? launch_one<256, 2, true>(p, grid, smem, stream)
                    : launch_one<0, 2, true>(p, grid, smem, stream);
    if (channels == 1 && i16)
        return h256 ? launch_one<256, 1, true>(p, grid, smem, stream)
                    : launch_one<0, 1, true>(p, grid, smem, stream);
    return fail(AF_ERR_UNSUPPORTED, "features: channels must be 1 or 2 (got %d)", channels);
}

}  // namespace feat
}  // namespace af
