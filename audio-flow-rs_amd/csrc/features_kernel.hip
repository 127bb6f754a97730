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
#include "features_dev.hpp"

namespace af {
namespace feat {

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

// ---------------------------------------------------------------------------
template <int HOP_T, int CH, bool I16, bool STAMPS>
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
    int* lmeta = reinterpret_cast<int*>(ttab + 32 * kTwRow);  // [4][kMaxSteps]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool aligned = p.aligned16 != 0;

    // work range of this workgroup, in 16-frame units
    const int64_t G = gridDim.x, b = blockIdx.x;
    int64_t u = p.total_units * b / G;
    const int64_t u_end = p.total_units * (b + 1) / G;
    if (u >= u_end) return;

    // window and twiddle tables -> LDS (made visible by the first barrier below)
    for (int i = tid; i < 32 * kWinRow / 4; i += kThreads)
        reinterpret_cast<float4*>(wtab)[i] = reinterpret_cast<const float4*>(p.wintab)[i];
    for (int i = tid; i < 32 * kTwRow / 4; i += kThreads)
        reinterpret_cast<float4*>(ttab)[i] = reinterpret_cast<const float4*>(p.twtab)[i];
    for (int i = tid; i < kWaves * kMaxSteps; i += kThreads) lmeta[i] = p.meta[i];
    // this wave's share of the mel plan
    const float* __restrict__ wfrag_w = p.wfrag + (((size_t)w * kMaxSlotsW) * 64 + lane) * 4;
    const int n_slots_w = p.plan.wave_nslots[w];
    int slot_end_w[kMaxSlotsW];
#pragma unroll
    for (int q = 0; q < kMaxSlotsW; ++q) slot_end_w[q] = p.plan.slot_end[w][q];
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
    // experimental stagger (AF_FEATURES_STAGGER): delay a subset of workgroups by about half a
    // unit so that co-resident workgroups alternate VALU-heavy and MFMA/barrier phases
    if (p.stagger_mode == 1 && (blockIdx.x & 1)) for (int i = 0; i < p.stagger_len; ++i) __builtin_amdgcn_s_sleep(64);
    if (p.stagger_mode == 2 && blockIdx.x >= gridDim.x / 2) for (int i = 0; i < p.stagger_len; ++i) __builtin_amdgcn_s_sleep(64);
    bool fresh = true;

    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = STAMPS ? __builtin_amdgcn_s_memtime() : 0;

    while (u < u_end) {
        while (p.unit_off[c + 1] <= u) ++c;       // skip clips without frames
        const int64_t c_pcm0 = p.pcm_off[c];
        const int64_t c_n = p.pcm_len[c];
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
        float wf[kMaxSteps];
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
                for (int q4 = 0; q4 < kMaxSlotsW; q4 += 4) {
                    if (q4 < n_slots_w) {         // wave-uniform: skip unused slots
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
                    St::convert(pf[i], m);
                    St::store(tile + tail_len + v * MS, m);
                }
            } else {
                for (int v = tid; v * MS < 16 * hop; v += kThreads) {
                    float m[MS];
                    St::load(p.pcm, c_pcm0, next_pos0 + (int64_t)v * MS, c_n, aligned, m);
                    St::store(tile + tail_len + v * MS, m);
                }
            }
        }

        AF_STAMP(4);
        {
            const int j = lane & 15;
            float* orow = p.mel + (c_out0 + F0 + j) * n_mels;
            const float* pbase = wbufs + (j >> 2) * kWbuf + (j & 3) * kProw + (lane >> 4);
            mel_phase<kMaxSteps, STAMPS>(pbase, lmeta + w * kMaxSteps, wf, n_slots_w, slot_end_w, orow, F0 + j < c_frames,
                              vec_ok, lane, acc_t, t_prev);
        }
        AF_STAMP(6);
        AF_STAMP(7);
        ++u;
        fresh = !has_next;
    }
    if (STAMPS && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(p.stamps + w * 8 + i, acc_t[i]);
}

size_t features_smem_bytes(int hop)
{
    const int tile_len = 15 * hop + kNfft;
    const int tile_alloc = (tile_len + 7) & ~7;
    return sizeof(float) * ((size_t)tile_alloc + kWaves * kWbuf + 32 * kWinRow + 32 * kTwRow +
                            kWaves * kMaxSteps);
}

template <int HOP_T, int CH, bool I16, bool STAMPS>
static af_status launch_k(const Params& p, int grid, size_t smem, hipStream_t stream)
{
    auto kern = af_features_kernel<HOP_T, CH, I16, STAMPS>;
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

template <int HOP_T, int CH, bool I16>
static af_status launch_one(const Params& p, int grid, size_t smem, hipStream_t stream)
{
    return p.stamps ? launch_k<HOP_T, CH, I16, true>(p, grid, smem, stream)
                    : launch_k<HOP_T, CH, I16, false>(p, grid, smem, stream);
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
