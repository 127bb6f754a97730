// pcm_kernels.hip -- the reference's small PCM ops as HBM-bound gfx950 kernels:
//   to_mono        modules/audio/capture.rs:30-42
//   PCM16 quantise modules/network/websocket.rs:246-251 (+ :254 base64)
//   VAD energy     modules/audio/vad.rs:157-168
// All are pure streaming kernels: 16-byte loads/stores per lane, a grid that covers the array
// (the grid-stride loops only run for more than 2^30 workgroups), no LDS except the VAD slice tile.
#include <cmath>
#include <type_traits>
#include <vector>

#include "af_common.hpp"
#include "pcm_dev.hpp"

namespace af {
namespace {

constexpr int kBlock = 256;

// One pass of the kernel's (unrolled) loop per thread: a grid that covers the whole array moves
// 6.4-6.6 TB/s through a 2:1 kernel of to_mono's shape where 2048 persistent workgroups looping over it reach
// 5.4-5.7 (scripts/ubench/stream_sweep.hip: block size, loads in flight and walk order matter far less).
inline int grid_for(size_t work_items, int per_thread = 1)
{
    const size_t per_wg = (size_t)kBlock * per_thread;
    size_t g = (work_items + per_wg - 1) / per_wg;
    if (g > (1u << 30)) g = 1u << 30;                    // the kernels loop over what is left
    if (g < 1) g = 1;
    return (int)g;
}

// ---- to_mono ---------------------------------------------------------------
// stereo fast path: one float4 (two frames) per lane and load, two loads in flight per lane
constexpr int kUnroll = 2;
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef unsigned vu2 __attribute__((ext_vector_type(2)));

__global__ void to_mono2_kernel(const float* __restrict__ in, size_t n_pairs /* = n_frames/2 */,
                                float* __restrict__ out)
{
    // every workgroup walks its own contiguous range, kUnroll * 4 KB at a time
    const vf4* src = reinterpret_cast<const vf4*>(in);
    vf2* dst = reinterpret_cast<vf2*>(out);
    const size_t per = (n_pairs + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per;
    const size_t hi = lo + per < n_pairs ? lo + per : n_pairs;
    size_t i = lo + threadIdx.x;
    for (; i + (kUnroll - 1) * kBlock < hi; i += kUnroll * kBlock) {
        vf4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = __builtin_nontemporal_load(src + i + u * kBlock);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            vf2 o;
            o.x = (v[u].x + v[u].y) / 2.0f;    // capture.rs:37-38: sum, then / channels
            o.y = (v[u].z + v[u].w) / 2.0f;
            __builtin_nontemporal_store(o, dst + i + u * kBlock);
        }
    }
    for (; i < hi; i += kBlock) {
        const vf4 v = src[i];
        vf2 o;
        o.x = (v.x + v.y) / 2.0f;
        o.y = (v.z + v.w) / 2.0f;
        dst[i] = o;
    }
}

// general channel count, including the trailing partial frame (`chunks`)
__global__ void to_mono_kernel(const float* __restrict__ in, size_t n_samples, unsigned channels,
                               size_t first_frame, size_t n_out, float* __restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t f = first_frame + (size_t)blockIdx.x * blockDim.x + threadIdx.x; f < n_out;
         f += stride) {
        const size_t b = f * channels;
        const size_t e = b + channels < n_samples ? b + channels : n_samples;
        // capture.rs:36 `chunk.iter().sum::<f32>()`: a left fold from the additive identity -0.0 (what `impl Sum for f32`
        // folds from in the toolchains edition 2024 needs; docs/OPS.md "to_mono"), i.e. x0 + x1 + ... with nothing in
        // front: a frame of all -0.0 stays -0.0, as in to_mono2_kernel's (L + R) / 2 and in every fused load
        float sum = -0.0f;
        for (size_t j = b; j < e; ++j) sum = sum + in[j];   // left fold, f32
        out[f] = sum / (float)channels;
    }
}

// ---- PCM16 quantiser ---------------------------------------------------------
// (quant1, the base64 primitives: pcm_dev.hpp, shared with the fused wire launch)
using pcmdev::quant1;
using pcmdev::b64_encode16;

__device__ __forceinline__ vu2 quant4(const vf4 v)
{
    const unsigned a = (unsigned)quant1(v.x) & 0xffffu, b = (unsigned)quant1(v.y) & 0xffffu;
    const unsigned c = (unsigned)quant1(v.z) & 0xffffu, d = (unsigned)quant1(v.w) & 0xffffu;
    vu2 o;
    o.x = a | (b << 16);                    // little-endian, websocket.rs:249
    o.y = c | (d << 16);
    return o;
}

__global__ void quantize_kernel(const float* __restrict__ in, size_t n, int16_t* __restrict__ out,
                                int vec_ok)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = vec_ok ? n / 4 : 0;
    const vf4* src = reinterpret_cast<const vf4*>(in);
    vu2* dst = reinterpret_cast<vu2*>(out);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (kUnroll - 1) * stride < n4; i += kUnroll * stride) {
        vf4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) __builtin_nontemporal_store(quant4(v[u]), dst + i + u * stride);
    }
    for (; i < n4; i += stride) dst[i] = quant4(src[i]);
    for (size_t j = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride)
        out[j] = (int16_t)quant1(in[j]);
}

// ---- quantise + base64 ---------------------------------------------------------
// 6 samples = 12 bytes -> 16 characters (one 16-byte store): pcmdev::b64_encode16
// Fast path: a lane takes 12 consecutive samples = three 16-byte loads (the three loads of a
// wave cover one contiguous 3 KB span, every cache line is used whole) -> two 16-byte stores.
// The last (n mod 12) samples, and everything when a pointer is not 16-byte aligned, go
// through the 6-sample path below.
__global__ __launch_bounds__(kBlock) void pcm16_base64_kernel(const float* __restrict__ in, size_t n,
                                                              char* __restrict__ out, int vec_ok)
{
    const size_t n_bytes = 2 * n;
    const size_t n12 = vec_ok ? n / 12 : 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = t0; i < n12; i += stride) {
        const float4* src = reinterpret_cast<const float4*>(in) + 3 * i;
        const float4 a = src[0], b = src[1], c = src[2];
        const float x0[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
        const float x1[6] = {b.z, b.w, c.x, c.y, c.z, c.w};
        uint4* dst = reinterpret_cast<uint4*>(out) + 2 * i;
        dst[0] = b64_encode16(x0);
        dst[1] = b64_encode16(x1);
    }
    // remaining 6-sample groups
    const size_t g0 = n12 * 2, n_groups = (n + 5) / 6;
    for (size_t g = g0 + t0; g < n_groups; g += stride) {
        float x[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) x[j] = g * 6 + j < n ? in[g * 6 + j] : 0.0f;
        const uint4 o = b64_encode16(x);
        const size_t byte0 = g * 12;
        if (byte0 + 12 <= n_bytes) {
            const unsigned w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int c = 0; c < 16; ++c) out[g * 16 + c] = (char)((w[c >> 2] >> (8 * (c & 3))) & 0xffu);
        } else {
            // last, partial group: whole 3-byte triples, then '=' padding
            pcmdev::b64_store_tail(out + g * 16, o, (unsigned)(n_bytes - byte0));      // 2, 4, 6, 8, 10
        }
    }
}

// ---- VAD energy ----------------------------------------------------------------
// frame_sweep (level meter): one wave64 per window, lanes stride it (16-byte loads when the
// windows are 16-byte aligned).
template <class F>
__device__ __forceinline__ void frame_sweep(const float* __restrict__ x, size_t len, int lane, bool vec, F&& f)
{
    size_t done = 0;
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const size_t n4 = len / 4;
        size_t i = lane;
        for (; i + 192 < n4; i += 256) {                  // four loads in flight, folded in index order
            const float4 a = x4[i], b = x4[i + 64], c = x4[i + 128], d = x4[i + 192];
            f(a.x); f(a.y); f(a.z); f(a.w);
            f(b.x); f(b.y); f(b.z); f(b.w);
            f(c.x); f(c.y); f(c.z); f(c.w);
            f(d.x); f(d.y); f(d.z); f(d.w);
        }
        for (; i < n4; i += 64) {
            const float4 v = x4[i];
            f(v.x); f(v.y); f(v.z); f(v.w);
        }
        done = n4 * 4;
    }
    for (size_t i = done + lane; i < len; i += 64) f(x[i]);
}

// The reference folds left in f32 (`frame.iter().map(|&x| x * x).sum()`, vad.rs:163-165: no FMA, one
// rounding per product and per add), so a tree sum differs in the last bits and a frame within
// rounding of the threshold could flip the VAD state.  Here ONE LANE folds ONE FRAME, in order:
// bit-identical to the reference.  A wave owns 64 consecutive frames and walks them in slices of 32
// samples: the slice is loaded coalesced (8 lanes x 16 B per frame run) into the wave's own LDS tile
// with a 33-dword row stride, then every lane reads its own row (conflict-free) and folds it.  The
// four waves of a workgroup never wait for each other (no s_barrier: a tile has one writer wave).
// (slices of 16 or 64 samples, or 128 frames per workgroup, measured 10-35 % slower)
constexpr int kVadFrames = 256, kVadSlice = 32, kVadStride = kVadSlice + 1;
__device__ __forceinline__ void vad_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(kVadFrames) void vad_energy_kernel(const float* __restrict__ in, size_t n_frames,
                                                                size_t frame_len, float* __restrict__ energy, int vec)
{
#pragma clang fp contract(off)
    __shared__ float tile[kVadFrames * kVadStride];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* wt = tile + w * 64 * kVadStride;
    for (size_t fb = (size_t)blockIdx.x * kVadFrames + 64 * w; fb < n_frames; fb += (size_t)gridDim.x * kVadFrames) {
        const int nf = (int)(n_frames - fb < (size_t)64 ? n_frames - fb : (size_t)64);   // frames of this wave
        const float* x = in + fb * frame_len;
        float s = 0.0f;
        for (size_t i0 = 0; i0 < frame_len; i0 += kVadSlice) {
            const int tcur = (int)(frame_len - i0 < (size_t)kVadSlice ? frame_len - i0 : (size_t)kVadSlice);
            vad_wave_sync();                                         // the previous slice has been folded
            if (vec && tcur == kVadSlice) {
                float4 v[kVadSlice / 4];
#pragma unroll
                for (int k = 0; k < kVadSlice / 4; ++k) {
                    const int q = lane + 64 * k, r = q / (kVadSlice / 4), part = q % (kVadSlice / 4);
                    if (r < nf) v[k] = *reinterpret_cast<const float4*>(x + (size_t)r * frame_len + i0 + 4 * part);
                }
#pragma unroll
                for (int k = 0; k < kVadSlice / 4; ++k) {
                    const int q = lane + 64 * k, r = q / (kVadSlice / 4), part = q % (kVadSlice / 4);
                    if (r < nf) {
                        float* d = wt + r * kVadStride + 4 * part;
                        d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
                    }
                }
            } else {
                for (int q = lane; q < nf * tcur; q += 64) {
                    const int r = q / tcur, e = q - r * tcur;
                    wt[r * kVadStride + e] = x[(size_t)r * frame_len + i0 + e];
                }
            }
            vad_wave_sync();
            if (lane < nf) {
                const float* row = wt + lane * kVadStride;
                if (tcur == kVadSlice) {
#pragma unroll
                    for (int e = 0; e < kVadSlice; ++e) { const float p = row[e] * row[e]; s = s + p; }
                } else {
                    for (int e = 0; e < tcur; ++e) { const float p = row[e] * row[e]; s = s + p; }
                }
            }
        }
        if (lane < nf) energy[fb + lane] = frame_len ? s / (float)frame_len : 0.0f;   // vad.rs:158,167
    }
}

// ---- level meter --------------------------------------------------------------
// AudioLevel{rms, peak, dbfs} per window (specs/0002-design.md:727-732; payload of the
// audio_level event, events/mod.rs:73-75,112-115).  One wave64 per window.
__global__ void level_meter_kernel(const float* __restrict__ in, size_t n_windows, size_t window_len,
                                   float* __restrict__ rms, float* __restrict__ peak,
                                   float* __restrict__ dbfs, int vec)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t f = wave; f < n_windows; f += n_waves) {
        float s = 0.0f, m = 0.0f;
        frame_sweep(in + f * window_len, window_len, lane, vec != 0, [&](float v) {
            s += v * v;
            m = fmaxf(m, fabsf(v));
        });
        // wave reduction on the DPP path (quad swaps, half-row and row mirrors: every lane of a row of 16
        // ends with the row's result), then the four rows through readlane -- no LDS round trips
        auto dpp = [](float v, auto ctrl) {
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value,
                                                                        0xF, 0xF, false));
        };
        auto step = [&](auto ctrl) {
            s += dpp(s, ctrl);
            m = fmaxf(m, dpp(m, ctrl));
        };
        step(std::integral_constant<int, 0xB1>{});       // quad_perm [1,0,3,2]
        step(std::integral_constant<int, 0x4E>{});       // quad_perm [2,3,0,1]
        step(std::integral_constant<int, 0x141>{});      // row_half_mirror
        step(std::integral_constant<int, 0x140>{});      // row_mirror
        auto row = [](float v, int l) {
            return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
        };
        s = (row(s, 0) + row(s, 16)) + (row(s, 32) + row(s, 48));
        m = fmaxf(fmaxf(row(m, 0), row(m, 16)), fmaxf(row(m, 32), row(m, 48)));
        if (lane == 0) {
            const float r = window_len ? sqrtf(s / (float)window_len) : 0.0f;
            rms[f] = r;
            peak[f] = m;
            dbfs[f] = r > 0.0f ? 20.0f * log10f(r) : -INFINITY;
        }
    }
}

}  // namespace
}  // namespace af

using namespace af;

extern "C" {

size_t af_to_mono_out_len(size_t n_samples, uint16_t channels)
{
    if (channels == 0) return 0;
    if (channels == 1) return n_samples;
    return (n_samples + channels - 1) / channels;
}

af_status af_to_mono_device(const float* in_dev, size_t n_samples, uint16_t channels,
                            float* out_dev, void* stream)
{
    AF_REQUIRE(channels > 0, "to_mono: channels must be > 0");
    if (n_samples == 0) return AF_OK;
    AF_REQUIRE(in_dev && out_dev, "to_mono: null buffer");
    hipStream_t s = as_stream(stream);
    const size_t n_out = af_to_mono_out_len(n_samples, channels);
    if (channels == 1) {                                   // capture.rs:31-33
        AF_HIP(hipMemcpyAsync(out_dev, in_dev, n_samples * sizeof(float), hipMemcpyDeviceToDevice, s));
        return AF_OK;
    }
    size_t done = 0;
    if (channels == 2 && reinterpret_cast<uintptr_t>(in_dev) % 16 == 0 &&
        reinterpret_cast<uintptr_t>(out_dev) % 8 == 0) {
        const size_t n_pairs = (n_samples / 2) / 2;        // whole frames, two per lane
        if (n_pairs) {
            hipLaunchKernelGGL(to_mono2_kernel, dim3(grid_for(n_pairs, kUnroll)), dim3(kBlock), 0, s, in_dev,
                               n_pairs, out_dev);
            done = n_pairs * 2;
        }
    }
    if (done < n_out)
        hipLaunchKernelGGL(to_mono_kernel, dim3(grid_for(n_out - done)), dim3(kBlock), 0, s, in_dev,
                           n_samples, (unsigned)channels, done, n_out, out_dev);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

af_status af_to_mono(const float* in, size_t n_samples, uint16_t channels, float* out, size_t out_cap,
                     size_t* n_out)
{
    AF_REQUIRE(channels > 0, "to_mono: channels must be > 0");
    const size_t no = af_to_mono_out_len(n_samples, channels);
    if (n_out) *n_out = no;
    if (out_cap < no) return fail(AF_ERR_BUFFER_TOO_SMALL, "to_mono: need %zu floats", no);
    if (no == 0) return AF_OK;
    AF_REQUIRE(in && out, "to_mono: null buffer");
    AF_TRY(ensure_device(-1));
    DevBuf din, dout;
    AF_TRY(din.reserve(n_samples * sizeof(float)));
    AF_TRY(dout.reserve(no * sizeof(float)));
    AF_HIP(hipMemcpy(din.p, in, n_samples * sizeof(float), hipMemcpyHostToDevice));
    AF_TRY(af_to_mono_device(din.as<float>(), n_samples, channels, dout.as<float>(), nullptr));
    AF_HIP(hipMemcpy(out, dout.p, no * sizeof(float), hipMemcpyDeviceToHost));
    return AF_OK;
}

af_status af_quantize_pcm16_device(const float* in_dev, size_t n, int16_t* out_dev, void* stream)
{
    if (n == 0) return AF_OK;
    AF_REQUIRE(in_dev && out_dev, "quantize: null buffer");
    const int vec_ok = reinterpret_cast<uintptr_t>(in_dev) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(out_dev) % 8 == 0;
    hipLaunchKernelGGL(quantize_kernel, dim3(grid_for((n + 3) / 4, kUnroll)), dim3(kBlock), 0, as_stream(stream),
                       in_dev, n, out_dev, vec_ok);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

af_status af_quantize_pcm16(const float* in, size_t n, int16_t* out)
{
    if (n == 0) return AF_OK;
    AF_REQUIRE(in && out, "quantize: null buffer");
    AF_TRY(ensure_device(-1));
    DevBuf din, dout;
    AF_TRY(din.reserve(n * sizeof(float)));
    AF_TRY(dout.reserve(n * sizeof(int16_t)));
    AF_HIP(hipMemcpy(din.p, in, n * sizeof(float), hipMemcpyHostToDevice));
    AF_TRY(af_quantize_pcm16_device(din.as<float>(), n, dout.as<int16_t>(), nullptr));
    AF_HIP(hipMemcpy(out, dout.p, n * sizeof(int16_t), hipMemcpyDeviceToHost));
    return AF_OK;
}

size_t af_base64_len(size_t n_bytes) { return 4 * ((n_bytes + 2) / 3); }

af_status af_pcm16_base64_device(const float* in_dev, size_t n, char* out_dev, void* stream)
{
    if (n == 0) return AF_OK;
    AF_REQUIRE(in_dev && out_dev, "pcm16_base64: null buffer");
    const int vec_ok = reinterpret_cast<uintptr_t>(out_dev) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(in_dev) % 16 == 0;
    hipLaunchKernelGGL(pcm16_base64_kernel, dim3(grid_for((n + 11) / 12)), dim3(kBlock), 0,
                       as_stream(stream), in_dev, n, out_dev, vec_ok);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

af_status af_pcm16_base64(const float* in, size_t n, char* out, size_t out_cap, size_t* n_out)
{
    const size_t len = af_base64_len(2 * n);
    if (n_out) *n_out = len;
    if (out_cap < len) return fail(AF_ERR_BUFFER_TOO_SMALL, "pcm16_base64: need %zu chars", len);
    if (n == 0) return AF_OK;
    AF_REQUIRE(in && out, "pcm16_base64: null buffer");
    AF_TRY(ensure_device(-1));
    DevBuf din, dout;
    AF_TRY(din.reserve(n * sizeof(float)));
    AF_TRY(dout.reserve(len + 16));
    AF_HIP(hipMemcpy(din.p, in, n * sizeof(float), hipMemcpyHostToDevice));
    AF_TRY(af_pcm16_base64_device(din.as<float>(), n, dout.as<char>(), nullptr));
    AF_HIP(hipMemcpy(out, dout.p, len, hipMemcpyDeviceToHost));
    return AF_OK;
}

af_status af_vad_energy_device(const float* in_dev, size_t n_frames, size_t frame_len,
                               float* energy_dev, void* stream)
{
    if (n_frames == 0) return AF_OK;
    AF_REQUIRE(energy_dev && (in_dev || frame_len == 0), "vad_energy: null buffer");
    size_t g = (n_frames + kVadFrames - 1) / kVadFrames;      // one lane per frame, 256 frames per workgroup
    if (g > (1u << 30)) g = 1u << 30;
    const int vec = frame_len % 4 == 0 && reinterpret_cast<uintptr_t>(in_dev) % 16 == 0;
    hipLaunchKernelGGL(vad_energy_kernel, dim3((int)g), dim3(kVadFrames), 0, as_stream(stream), in_dev,
                       n_frames, frame_len, energy_dev, vec);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

af_status af_vad_energy(const float* in, size_t n_frames, size_t frame_len, float* energy_out)
{
    if (n_frames == 0) return AF_OK;
    AF_REQUIRE(energy_out && (in || frame_len == 0), "vad_energy: null buffer");
    AF_TRY(ensure_device(-1));
    DevBuf din, dout;
    const size_t n = n_frames * frame_len;
    AF_TRY(din.reserve(n * sizeof(float)));
    AF_TRY(dout.reserve(n_frames * sizeof(float)));
    if (n) AF_HIP(hipMemcpy(din.p, in, n * sizeof(float), hipMemcpyHostToDevice));
    AF_TRY(af_vad_energy_device(din.as<float>(), n_frames, frame_len, dout.as<float>(), nullptr));
    AF_HIP(hipMemcpy(energy_out, dout.p, n_frames * sizeof(float), hipMemcpyDeviceToHost));
    return AF_OK;
}

}  // extern "C"

extern "C" {

af_status af_level_meter_device(const float* in_dev, size_t n_windows, size_t window_len,
                                float* rms_dev, float* peak_dev, float* dbfs_dev, void* stream)
{
    if (n_windows == 0) return AF_OK;
    AF_REQUIRE(rms_dev && peak_dev && dbfs_dev && (in_dev || window_len == 0), "level_meter: null buffer");
    size_t g = (n_windows * 64 + kBlock - 1) / kBlock;
    if (g > (1u << 30)) g = 1u << 30;
    const int vec = window_len % 4 == 0 && reinterpret_cast<uintptr_t>(in_dev) % 16 == 0;
    hipLaunchKernelGGL(level_meter_kernel, dim3((int)g), dim3(kBlock), 0, as_stream(stream), in_dev,
                       n_windows, window_len, rms_dev, peak_dev, dbfs_dev, vec);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

af_status af_level_meter(const float* in, size_t n_windows, size_t window_len, float* rms,
                         float* peak, float* dbfs)
{
    if (n_windows == 0) return AF_OK;
    AF_REQUIRE(rms && peak && dbfs && (in || window_len == 0), "level_meter: null buffer");
    AF_TRY(ensure_device(-1));
    DevBuf din, dout;
    const size_t n = n_windows * window_len;
    AF_TRY(din.reserve(n * sizeof(float)));
    AF_TRY(dout.reserve(3 * n_windows * sizeof(float)));
    if (n) AF_HIP(hipMemcpy(din.p, in, n * sizeof(float), hipMemcpyHostToDevice));
    float* o = dout.as<float>();
    AF_TRY(af_level_meter_device(din.as<float>(), n_windows, window_len, o, o + n_windows,
                                 o + 2 * n_windows, nullptr));
    AF_HIP(hipMemcpy(rms, o, n_windows * sizeof(float), hipMemcpyDeviceToHost));
    AF_HIP(hipMemcpy(peak, o + n_windows, n_windows * sizeof(float), hipMemcpyDeviceToHost));
    AF_HIP(hipMemcpy(dbfs, o + 2 * n_windows, n_windows * sizeof(float), hipMemcpyDeviceToHost));
    return AF_OK;
}

}  // extern "C"
