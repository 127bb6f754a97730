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
//     non-zero band of each group of <= 16 filters is multiplied (about 134 K-steps
//     of 4 bins instead of 645 dense); whole groups are dealt to the 4 waves by the
//     host (features_api.hip, build_plan), so every wave stores its own 16-byte quads
//     of mel values straight to HBM and nothing is reduced across waves.
//   * other outputs of the same kernel (Params flags / template SEP): the power rows
//     themselves (AF_OUT_POWER), log10 of the mel values, one spectrogram per channel.
#include "features_dev.hpp"
#include "fft_regs.hpp"

namespace af {
namespace feat {

// ---------------------------------------------------------------------------
// FFT phase of one wave: 4 frames (two per half-wave) -> 4 power rows in LDS.
// Part A: window, first radix-32, twiddles, transpose through LDS -> ur/ui hold the
// second stage's inputs.  Part B: second radix-32, spectrum split, power rows.
template <int HOP_T>
__device__ __forceinline__ void fft_phase_a(const float* __restrict__ tile, float* wbuf, int hop,
                                            int w, int lane, const float* __restrict__ wtab,
                                            const float* __restrict__ ttab, float (&ur)[32],
                                            float (&ui)[32])
{
    const int half = lane >> 5, n2 = lane & 31;
    const int fa = 4 * w + 2 * half;              // first frame of this half's pair
    float re[32], im[32];
    {
        // window column of this lane: 0.5 * hann[32 n1 + n2], n1 = 0..31 (8 x ds_read_b128)
        float win[32];
        const float4* wrow = reinterpret_cast<const float4*>(wtab + n2 * kWinRow);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 v = wrow[j];
            win[4 * j] = v.x; win[4 * j + 1] = v.y; win[4 * j + 2] = v.z; win[4 * j + 3] = v.w;
        }
        const float* src = tile + fa * hop + n2;
        if constexpr (HOP_T != 0 && HOP_T % 32 == 0) {
            // frame b = frame a shifted by HOP_T samples = SH registers
            constexpr int SH = HOP_T / 32;
            float s[32 + SH];
#pragma unroll
            for (int j = 0; j < 32 + SH; ++j) s[j] = src[32 * j];
#pragma unroll
            for (int e = 0; e < 32; e += 2) {      // window fused into the twiddle-free first stage
                const int na = brev5(e), nb = na + 16;
                const float mr = s[nb] * win[nb], mi = s[nb + SH] * win[nb];
                re[e] = __builtin_fmaf(s[na], win[na], mr);
                re[e + 1] = __builtin_fmaf(s[na], win[na], -mr);
                im[e] = __builtin_fmaf(s[na + SH], win[na], mi);
                im[e + 1] = __builtin_fmaf(s[na + SH], win[na], -mi);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 32; e += 2) {
                const int na = brev5(e), nb = na + 16;
                const float mr = src[32 * nb] * win[nb], mi = src[hop + 32 * nb] * win[nb];
                re[e] = __builtin_fmaf(src[32 * na], win[na], mr);
                re[e + 1] = __builtin_fmaf(src[32 * na], win[na], -mr);
                im[e] = __builtin_fmaf(src[hop + 32 * na], win[na], mi);
                im[e + 1] = __builtin_fmaf(src[hop + 32 * na], win[na], -mi);
            }
        }
    }

    fftr::dit_fft<32, 2>(re, im);                 // over n1 (bit-reversed on entry) -> Y[k1] at k1

    // inter-stage twiddle e^{-2 pi i n2 k1 / 1024}; row n2 of the LDS table holds
    // (cos, sin) for k1 = 0..31, read as 8-byte pairs
    {
        const float2* trow = reinterpret_cast<const float2*>(ttab + n2 * kTwRow);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j > 0) {
                const float2 t = trow[2 * j];
                const int r = 2 * j;
                const float xr = re[r], xi = im[r];
                re[r] = xr * t.x + xi * t.y;
                im[r] = xi * t.x - xr * t.y;
            }
            const float2 t = trow[2 * j + 1];
            const int r = 2 * j + 1;
            const float xr = re[r], xi = im[r];
            re[r] = xr * t.x + xi * t.y;
            im[r] = xi * t.x - xr * t.y;
        }
    }

    // transpose (k1, n2) through LDS, real parts then imaginary parts
    float* xb = wbuf + half * kHalfDw;
#pragma unroll
    for (int k1 = 0; k1 < 32; ++k1) xb[k1 * kXStride + n2] = re[k1];
    wave_sync();
#pragma unroll
    for (int n = 0; n < 32; ++n) ur[brev5(n)] = xb[n2 * kXStride + n];
    wave_sync();
#pragma unroll
    for (int k1 = 0; k1 < 32; ++k1) xb[k1 * kXStride + n2] = im[k1];
    wave_sync();
#pragma unroll
    for (int n = 0; n < 32; ++n) ui[brev5(n)] = xb[n2 * kXStride + n];
    wave_sync();

}

__device__ __forceinline__ void fft_phase_b(float* wbuf, int lane, float (&ur)[32], float (&ui)[32])
{
    const int half = lane >> 5, n2 = lane & 31;
    float* xb = wbuf + half * kHalfDw;
    fftr::dit_fft<32>(ur, ui);                    // over n2 -> Z[k1 + 32 k2] at k2

    // Split the packed spectrum: A[k] = (Z[k] + conj Z[N-k])/2 (frame a),
    // B[k] = (Z[k] - conj Z[N-k])/(2i) (frame b); the 1/2 is folded into the
    // window table.  Lane k1 needs Z[N-k] = lane (32-k1)%32, k2' = 31-q
    // (lane 0: k2' = (32-q)%32).  Publish k2 in {0,16..31} -> slots {0,1..16}.
    const int k1 = n2;
    const int d = (k1 == 0) ? 1 : 0;
    // row k1 of the exchange buffer: re of k2 in {0, 16..31} at slots 0..16, im at 17..33
    float* mine = xb + k1 * kPStride;
    const float* theirs = xb + ((32 - k1) & 31) * kPStride + d;
    const int first = d ? 0 : 16;                 // slot of the q = 0 partner

    mine[0] = ur[0];
    mine[17] = ui[0];
#pragma unroll
    for (int k2 = 16; k2 < 32; ++k2) {
        mine[k2 - 15] = ur[k2];
        mine[k2 + 2] = ui[k2];
    }
    wave_sync();
    // partner values are fetched a few q at a time and turned into power right away (the
    // power takes over the registers of Z[k]); the rows alias the exchange buffer, so they are
    // written only after every lane of the wave has read its partners
    float pa[17], pb[17];
    constexpr int kChunk = 6;
#pragma unroll
    for (int q0 = 0; q0 <= 16; q0 += kChunk) {
        float pr[kChunk], pi[kChunk];
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            const int q = q0 + j;
            if (q > 16) break;
            if (q == 0) {
                pr[j] = xb[((32 - k1) & 31) * kPStride + first];
                pi[j] = xb[((32 - k1) & 31) * kPStride + first + 17];
            } else {
                pr[j] = theirs[16 - q];
                pi[j] = theirs[16 - q + 17];
            }
        }
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            const int q = q0 + j;
            if (q > 16) break;
            const float zr = ur[q], zi = ui[q];
            const float a1 = zr + pr[j], a2 = zi - pi[j];
            const float b1 = zi + pi[j], b2 = zr - pr[j];
            pa[q] = a1 * a1 + a2 * a2;
            pb[q] = b1 * b1 + b2 * b2;
        }
    }
    wave_sync();

    float* prow_a = wbuf + (2 * half) * kProw + k1;
    float* prow_b = prow_a + kProw;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        prow_a[32 * q] = pa[q];
        prow_b[32 * q] = pb[q];
    }
    if (k1 == 0) {                                // Nyquist bin 512
        prow_a[512] = pa[16];
        prow_b[512] = pb[16];
    }
}

// Diagnostics (Params::stamps != null only): per-phase cycle sums of every wave.
#define AF_STAMP(i)                                                         \
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

// Table entries of clip c with scalar loads (the tables are read-only for the kernel and c is
// wave-uniform).  The wait is inside: hipcc does not track loads issued from inline asm.
struct ClipEntry {
    int64_t pcm0, n, out0, out1, u0, u1;
};
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ClipEntry load_clip_entry(const int64_t* pcm_off, const int64_t* pcm_len,
                                                     const int64_t* out_off, const int64_t* unit_off,
                                                     int c)
{
    const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane(c) * 8u;
    unsigned long long a, b;
    u32x4 o, u;
    asm volatile(
        "s_load_dwordx2 %0, %4, %8\n\t"
        "s_load_dwordx2 %1, %5, %8\n\t"
        "s_load_dwordx4 %2, %6, %8\n\t"
        "s_load_dwordx4 %3, %7, %8\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(a), "=&s"(b), "=&s"(o), "=&s"(u)
        : "s"(pcm_off), "s"(pcm_len), "s"(out_off), "s"(unit_off), "s"(off)
        : "memory");
    ClipEntry e;
    e.pcm0 = (int64_t)a;
    e.n = (int64_t)b;
    e.out0 = (int64_t)(((unsigned long long)o.y << 32) | o.x);
    e.out1 = (int64_t)(((unsigned long long)o.w << 32) | o.z);
    e.u0 = (int64_t)(((unsigned long long)u.y << 32) | u.x);
    e.u1 = (int64_t)(((unsigned long long)u.w << 32) | u.z);
    return e;
}

// ---------------------------------------------------------------------------
// SEP (stereo only): no downmix, table entry e is channel e & 1 of clip e / 2
template <int HOP_T, int CH, bool I16, bool STAMPS, bool SEP, int NSLOT>
__global__ __launch_bounds__(kThreads, 2) void af_features_kernel(const Params p)
{
    using St = Stage<CH, I16>;
    constexpr int MS = St::MS;
    // prefetch units per thread and iteration (compile-time only for HOP_T != 0)
    constexpr int PFU = HOP_T ? (16 * HOP_T / MS + kThreads - 1) / kThreads : 1;
    constexpr bool kPrefetch = HOP_T != 0 && PFU <= 8 && (16 * HOP_T / MS) % kThreads == 0;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int hop = HOP_T ? HOP_T : p.hop;
    const int tile_len = 15 * hop + kNfft;
    const int tile_alloc = (tile_len + 7) & ~7;
    const int tail_len = kNfft - hop;             // samples shared with the next unit
    float* tile = smem;
    float* wbufs = smem + tile_alloc;
    float* wtab = wbufs + kWaves * kWbuf;                     // [32][kWinRow]
    float* ttab = wtab + 32 * kWinRow;                        // [32][kTwRow]
    int* lmeta = reinterpret_cast<int*>(ttab + 32 * kTwRow);  // [4][kMaxSlotsW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool aligned = p.aligned16 != 0;

    // work range of this workgroup, in 16-frame units
    const int64_t G = gridDim.x, b = blockIdx.x;
    int64_t u = p.unit_begin + p.total_units * b / G;
    const int64_t u_end = p.unit_begin + p.total_units * (b + 1) / G;
    if (u >= u_end) return;

    // window and twiddle tables -> LDS (made visible by the first barrier below)
    for (int i = tid; i < 32 * kWinRow / 4; i += kThreads)
        reinterpret_cast<float4*>(wtab)[i] = reinterpret_cast<const float4*>(p.wintab)[i];
    for (int i = tid; i < 32 * kTwRow / 4; i += kThreads)
        reinterpret_cast<float4*>(ttab)[i] = reinterpret_cast<const float4*>(p.twtab)[i];
    for (int i = tid; i < kWaves * kMaxSlotsW; i += kThreads) lmeta[i] = p.meta[i];
    // this wave's share of the mel plan
    const float* __restrict__ wfrag_w = p.wfrag + (((size_t)w * kMaxSlotsW) * 64 + lane) * 4;
    const int n_slots_w = p.plan.wave_nslots[w];
    int slot_end_w[NSLOT];
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) slot_end_w[q] = p.plan.slot_end[w][q];
    const int n_mels = p.plan.n_mels;
    const bool vec_ok = (n_mels % 4 == 0) && (reinterpret_cast<uintptr_t>(p.mel) % 16 == 0);

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
    bool fresh = true;

    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = STAMPS ? __builtin_amdgcn_s_memtime() : 0;

    while (u < u_end) {
        // The clip's table entries come through the scalar cache: as vector loads they would make
        // every unit wait for vmcnt(0), i.e. for the previous unit's mel stores to reach memory.
        ClipEntry ce;
        for (;;) {
            ce = load_clip_entry(p.pcm_off, p.pcm_len, p.out_off, p.unit_off, c);
            if (ce.u1 > u) break;                 // skip clips without frames
            ++c;
        }
        const int64_t c_pcm0 = ce.pcm0;
        const int64_t c_n = ce.n;
        const int64_t c_out0 = ce.out0;
        const int64_t c_frames = ce.out1 - ce.out0;
        const int64_t c_u1 = ce.u1;
        const int64_t F0 = (u - ce.u0) * kFramesPerIter;
        const int sel = SEP ? (c & 1) : -1;       // entry = 2 clip + channel

        if (fresh) {
            __syncthreads();                      // nobody still reads the old tile
            const int64_t pos0 = F0 * hop;
            for (int v = tid; v * MS < tile_alloc; v += kThreads) {
                float m[MS];
                St::load(p.pcm, c_pcm0, pos0 + (int64_t)v * MS, c_n, aligned, m, sel);
                St::store(tile + v * MS, m);
            }
        }
        __syncthreads();                          // B1: tile ready
        AF_STAMP(0);

        const bool has_next = (u + 1 < u_end) && (u + 1 < c_u1);
        const int64_t next_pos0 = (F0 + kFramesPerIter) * hop + tail_len;  // first new sample
        uint4 pf[PFU];
        float tailr[4];
        if (has_next) {
            if constexpr (kPrefetch) {
                if (aligned && next_pos0 + 16 * hop <= c_n) {
                    // common case (wave-uniform): the whole new region lies inside the clip
                    const char* src = static_cast<const char*>(p.pcm) +
                                      (c_pcm0 + next_pos0) * St::kBytesPerFrame;
#pragma unroll
                    for (int i = 0; i < PFU; ++i)
                        pf[i] = *reinterpret_cast<const uint4*>(src + (unsigned)(tid + i * kThreads) * 16u);
                } else {
#pragma unroll
                    for (int i = 0; i < PFU; ++i) {
                        const int v = tid + i * kThreads;
                        pf[i] = St::load_raw(p.pcm, c_pcm0, next_pos0 + (int64_t)v * MS, c_n, aligned);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = tid + i * kThreads;
                tailr[i] = (j < tail_len) ? tile[16 * hop + j] : 0.0f;
            }
        }

        AF_STAMP(1);
        float wf[NSLOT * kSlotSteps];
        {
            float ur[32], ui[32];
            fft_phase_a<HOP_T>(tile, wbufs + w * kWbuf, hop, w, lane, wtab, ttab, ur, ui);
            // This wave's filterbank fragments (L2-resident, one float4 per slot).  A global
            // load takes several microseconds to come back while the chip streams PCM, so
            // they are issued here -- register pressure is lowest between the two radix-32
            // stages -- and consumed in the mel phase.
            {
                int opaque_zero;                  // pins the loads here (and defeats hoisting)
                asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
                const float4* wp = reinterpret_cast<const float4*>(wfrag_w) + opaque_zero;
#pragma unroll
                for (int q4 = 0; q4 < NSLOT; q4 += 4) {
                    if (q4 < n_slots_w && !p.out_power) {   // wave-uniform: skip unused slots
#pragma unroll
                        for (int q = q4; q < q4 + 4; ++q) {
                            const float4 v = wp[q * 64];
                            wf[4 * q] = v.x; wf[4 * q + 1] = v.y; wf[4 * q + 2] = v.z; wf[4 * q + 3] = v.w;
                        }
                    } else {
#pragma unroll
                        for (int s = 4 * q4; s < 4 * q4 + 16; ++s) wf[s] = 0.0f;
                    }
                }
            }
            fft_phase_b(wbufs + w * kWbuf, lane, ur, ui);
        }
        AF_STAMP(2);

        __syncthreads();                          // B2: power rows ready, tile reads done
        AF_STAMP(3);

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
                    float m[MS];
                    St::convert(pf[i], m, sel);
                    St::store(tile + tail_len + v * MS, m);
                }
            } else {
                for (int v = tid; v * MS < 16 * hop; v += kThreads) {
                    float m[MS];
                    St::load(p.pcm, c_pcm0, next_pos0 + (int64_t)v * MS, c_n, aligned, m, sel);
                    St::store(tile + tail_len + v * MS, m);
                }
            }
        }

        AF_STAMP(4);
        {
            const int j = lane & 15;
            float* orow = p.mel + (c_out0 + F0 + j) * n_mels;
            if (!p.out_power) {
                const int mf = lane & 15;
                mel_phase<STAMPS, NSLOT>(wbufs + (mf >> 2) * kWbuf + (mf & 3) * kProw + (lane >> 4),
                                         lmeta + w * kMaxSlotsW, wf, n_slots_w, slot_end_w, orow,
                                  F0 + j < c_frames, vec_ok, p.log_mel != 0, lane, acc_t, t_prev);
            } else {
                // power output: this wave's four rows [frame][513], consecutive lanes on
                // consecutive bins
                const float* rows = wbufs + w * kWbuf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t f = F0 + 4 * w + r;
                    if (f < c_frames) {
                        float* o = p.mel + (c_out0 + f) * kBins;
                        for (int k = lane; k < kBins; k += 64) o[k] = rows[r * kProw + k];
                    }
                }
            }
        }
        AF_STAMP(6);
        AF_STAMP(7);
        ++u;
        fresh = !has_next;
    }
    if (STAMPS && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(p.stamps + w * 8 + i, acc_t[i]);
}

// ---------------------------------------------------------------------------
// Stand-alone mel contraction (af_mel_apply): 16 power rows per iteration are staged in the
// layout the fused kernel's FFT leaves them in, then the same mel phase runs.  The filterbank
// fragments stay in registers for the whole launch.
__global__ __launch_bounds__(kThreads, 2) void af_mel_apply_kernel(const MelApplyParams p)
{
    __shared__ __attribute__((aligned(16))) float wbufs[kWaves * kWbuf];
    __shared__ int lmeta[kWaves * kMaxSlotsW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kWaves * kMaxSlotsW; i += kThreads) lmeta[i] = p.meta[i];
    const float4* wp = reinterpret_cast<const float4*>(p.wfrag + (((size_t)w * kMaxSlotsW) * 64 + lane) * 4);
    float wf[kMaxSteps];
#pragma unroll
    for (int q = 0; q < kMaxSlotsW; ++q) {
        const float4 v = wp[q * 64];
        wf[4 * q] = v.x; wf[4 * q + 1] = v.y; wf[4 * q + 2] = v.z; wf[4 * q + 3] = v.w;
    }
    const int n_slots_w = p.plan.wave_nslots[w];
    int slot_end_w[kMaxSlotsW];
#pragma unroll
    for (int q = 0; q < kMaxSlotsW; ++q) slot_end_w[q] = p.plan.slot_end[w][q];
    const int n_mels = p.plan.n_mels;
    const bool vec_ok = (n_mels % 4 == 0) && (reinterpret_cast<uintptr_t>(p.mel) % 16 == 0);
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
    const int64_t n_units = (p.n_frames + kFramesPerIter - 1) / kFramesPerIter;
    // wave w stages rows 4w .. 4w+3 of a unit: lane + 64 j walks the 513 bins (j = 8 is bin 512,
    // lane 0 only); the next unit's rows are fetched into registers under the MFMAs
    constexpr int kPer = 9;
    float pf[4 * kPer];
    float* my_rows = wbufs + w * kWbuf;
    auto fetch = [&](int64_t unit) {
        const int64_t F0 = unit * kFramesPerIter + 4 * w;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = F0 + r < p.n_frames;
            const float* src = p.power + (F0 + r) * kBins + lane;
#pragma unroll
            for (int j = 0; j < kPer; ++j)
                pf[r * kPer + j] = (ok && (j < 8 || lane == 0)) ? src[64 * j] : 0.0f;
        }
    };
    int64_t u = blockIdx.x;
    if (u < n_units) fetch(u);
    for (; u < n_units; u += gridDim.x) {
        const int64_t F0 = u * kFramesPerIter;
        __syncthreads();                              // the previous unit's operands are read
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < kPer; ++j)
                if (j < 8 || lane == 0) my_rows[r * kProw + lane + 64 * j] = pf[r * kPer + j];
        __syncthreads();
        if (u + gridDim.x < n_units) fetch(u + gridDim.x);
        const int j = lane & 15;
        float* orow = p.mel + (F0 + j) * n_mels;
        mel_phase<false, kMaxSlotsW>(wbufs + (j >> 2) * kWbuf + (j & 3) * kProw + (lane >> 4),
                                     lmeta + w * kMaxSlotsW, wf, n_slots_w, slot_end_w, orow,
                         F0 + j < p.n_frames, vec_ok, p.log_mel != 0, lane, acc_t, t_prev);
    }
}

af_status launch_mel_apply(const MelApplyParams& p, int grid, hipStream_t stream)
{
    if (grid <= 0) return AF_OK;
    hipLaunchKernelGGL(af_mel_apply_kernel, dim3(grid), dim3(kThreads), 0, stream, p);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

size_t features_smem_bytes(int hop)
{
    const int tile_len = 15 * hop + kNfft;
    const int tile_alloc = (tile_len + 7) & ~7;
    return sizeof(float) * ((size_t)tile_alloc + kWaves * kWbuf + 32 * kWinRow + 32 * kTwRow +
                            kWaves * kMaxSlotsW);
}

template <int HOP_T, int CH, bool I16, bool STAMPS, bool SEP, int NSLOT>
static af_status launch_k(const Params& p, int grid, size_t smem, hipStream_t stream)
{
    auto kern = af_features_kernel<HOP_T, CH, I16, STAMPS, SEP, NSLOT>;
    static LdsAttrFlags attr_set;
    AF_HIP(allow_big_lds(kern, 160 * 1024, attr_set));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), smem, stream, p);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

template <int HOP_T, int CH, bool I16>
static af_status launch_one(const Params& p, int grid, size_t smem, hipStream_t stream)
{
    if constexpr (CH == 2) {
        if (p.separate) return launch_k<HOP_T, CH, I16, false, true, kMaxSlotsW>(p, grid, smem, stream);
    }
    if (p.stamps) return launch_k<HOP_T, CH, I16, true, false, kMaxSlotsW>(p, grid, smem, stream);
    // the common filterbanks (80 mels) need at most 12 slots on a wave: 16 fewer live registers
    // in the FFT and 16 fewer operand registers in the mel phase
    int most = 0;
    for (int w = 0; w < kWaves; ++w) most = p.plan.wave_nslots[w] > most ? p.plan.wave_nslots[w] : most;
    if constexpr (HOP_T == 256) {
        if (most <= 12) return launch_k<HOP_T, CH, I16, false, false, 12>(p, grid, smem, stream);
    }
    return launch_k<HOP_T, CH, I16, false, false, kMaxSlotsW>(p, grid, smem, stream);
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
