// features_dev.hpp -- device code shared by the fused feature kernels (features_kernel.hip: the
// general kernel, features_v2.hip: the hop-256 kernel): PCM staging (to_mono, capture.rs:30-42),
// the banded mel contraction on MFMA, small helpers.
#pragma once

#include "features.hpp"

namespace af {
namespace feat {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ constexpr int brev5(int x)
{
    return ((x & 1) << 4) | ((x & 2) << 2) | (x & 4) | ((x & 8) >> 2) | ((x & 16) >> 4);
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

    // the 16 raw bytes holding mono samples [pos, pos+MS) of a clip whose first
    // frame is `base` and which has `n` frames; zero beyond the end of the clip.
    // No arithmetic here, so a prefetch does not have to wait for its data.
    __device__ static __forceinline__ uint4 load_raw(const void* pcm, int64_t base, int64_t pos,
                                                     int64_t n, bool aligned)
    {
        if (aligned && pos + MS <= n)
            return *reinterpret_cast<const uint4*>(static_cast<const char*>(pcm) +
                                                   (base + pos) * kBytesPerFrame);
        uint32_t r[4] = {0u, 0u, 0u, 0u};
        constexpr int E = I16 ? 8 : 4;                 // elements in 16 bytes
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (pos + e / CH < n) {
                const int64_t idx = (base + pos) * CH + e;
                if constexpr (I16)
                    r[e >> 1] |= (uint32_t)static_cast<const uint16_t*>(pcm)[idx] << (16 * (e & 1));
                else
                    r[e] = static_cast<const uint32_t*>(pcm)[idx];
            }
        }
        return uint4{r[0], r[1], r[2], r[3]};
    }

    // raw bytes -> mono samples.  sel < 0: to_mono downmix; sel = 0 / 1 (stereo only): that
    // channel alone ("separate channels" mode, wave-uniform).
    __device__ static __forceinline__ void convert(const uint4 raw, float (&m)[MS], int sel)
    {
        const uint32_t r[4] = {raw.x, raw.y, raw.z, raw.w};
        if constexpr (!I16) {
            float f[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(r[i]);
            if constexpr (CH == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i] = f[i];
            } else if (sel < 0) {
                m[0] = (f[0] + f[1]) * 0.5f;
                m[1] = (f[2] + f[3]) * 0.5f;
            } else {
                m[0] = sel ? f[1] : f[0];
                m[1] = sel ? f[3] : f[2];
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
            } else if (sel < 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i] = (f[2 * i] + f[2 * i + 1]) * 0.5f;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i] = sel ? f[2 * i + 1] : f[2 * i];
            }
        }
    }

    __device__ static __forceinline__ void load(const void* pcm, int64_t base, int64_t pos,
                                                int64_t n, bool aligned, float (&m)[MS], int sel)
    {
        convert(load_raw(pcm, base, pos, n, aligned), m, sel);
    }

    __device__ static __forceinline__ void store(float* dst, const float (&m)[MS])
    {
#pragma unroll
        for (int i = 0; i < MS; ++i) dst[i] = m[i];
    }
};

// ---------------------------------------------------------------------------
// mel phase: this wave's filter groups (pbase: this lane's power row + its k-group, i.e. frame
// lane % 16, bin offset lane / 16), banded W (A operand, in registers) times
// P^T (B operand, bins x 16 frames from LDS) on v_mfma_f32_16x16x4_f32.  All B
// operands of the unit are fetched up front so the MFMA chain never waits on LDS.
// Steps come in slots of 8 (branch-free); when a slot ends a group, the 16 frames
// x <=16 filters accumulator goes straight to HBM: lane (frame j, quarter kq)
// owns filters 4kq..4kq+3 of the group = one 16-byte store.
// MS = slots per wave compiled in (12 covers the 80-filter banks, 16 every supported one)
template <bool STAMPS, int MS>
__device__ __forceinline__ void mel_phase(const float* pbase, const int* __restrict__ lmeta,
                                          const float (&wf)[MS * kSlotSteps], int n_slots,
                                          const int (&slot_end)[MS],
                                          float* __restrict__ orow, bool fvalid, bool vec_ok,
                                          bool log_out, int lane, unsigned long long (&acc_t)[8],
                                          unsigned long long& t_prev)
{
    const int kq = lane >> 4;
    // Branch-free on purpose.  The 4 steps of a slot read consecutive bins (4 apart), so one
    // broadcast read per slot gives its first bin and the operands follow at fixed offsets
    // (padding steps read real bins of the same frame and multiply them by zero weights).
    float b[MS * kSlotSteps];
    {
        int meta[MS];
#pragma unroll
        for (int q = 0; q < MS; ++q) meta[q] = lmeta[q];
#pragma unroll
        for (int q = 0; q < MS; ++q) {
            const float* bq = pbase + meta[q];
#pragma unroll
            for (int i = 0; i < kSlotSteps; ++i) b[q * kSlotSteps + i] = bq[4 * i];
        }
    }
    if constexpr (STAMPS) {                       // diagnostics: operands landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        acc_t[5] += t - t_prev;
        t_prev = t;
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < MS; ++q) {
        if (q < n_slots) {
#pragma unroll
            for (int s = q * kSlotSteps; s < (q + 1) * kSlotSteps; s += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], b[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s + 1], b[s + 1], acc1, 0, 0, 0);
            }
            const int e = slot_end[q];            // wave-uniform (kernarg)
            if (e != 0) {
                f32x4 r = acc0 + acc1;
                if (log_out) {                    // wave-uniform: log10(max(mel, 1e-10))
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[i] = __log10f(fmaxf(r[i], 1e-10f));
                }
                acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
                acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
                const int m0 = e & 255, cnt = (e >> 8) & 31;
                const int i0 = 4 * kq;            // first filter of this lane in the group
                float* o = orow + m0 + i0;
                if (vec_ok && (cnt & 3) == 0) {   // wave-uniform choice
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


}  // namespace feat
}  // namespace af
