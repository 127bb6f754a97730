// pcm_kernels.hip -- the reference's small PCM ops as HBM-bound gfx950 kernels:
//   to_mono        modules/audio/capture.rs:30-42
//   PCM16 quantise modules/network/websocket.rs:246-251 (+ :254 base64)
//   VAD energy     modules/audio/vad.rs:157-168
// All are pure streaming kernels: 16-byte loads/stores per lane, grid-stride
// over at most 2048 workgroups, no LDS except the wave reduction of the energy.
#include <cmath>
#include <vector>

#include "af_common.hpp"

namespace af {
namespace {

constexpr int kBlock = 256;

inline int grid_for(size_t work_items)
{
    size_t g = (work_items + kBlock - 1) / kBlock;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- to_mono ---------------------------------------------------------------
// stereo fast path: one float4 (two frames) per lane and load, four loads in flight per lane
// (a lane that keeps a single 16-byte load in flight leaves HBM a third idle)
constexpr int kUnroll = 4;
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
        float sum = 0.0f;
        for (size_t j = b; j < e; ++j) sum = sum + in[j];   // left fold, f32
        out[f] = sum / (float)channels;
    }
}

// ---- PCM16 quantiser ---------------------------------------------------------
__device__ __forceinline__ int quant1(float x)
{
    // websocket.rs:248: (x.clamp(-1.0, 1.0) * 32767.0) as i16
    if (x != x) return 0;                   // NaN survives clamp, `as i16` gives 0
    x = x < -1.0f ? -1.0f : x;
    x = x > 1.0f ? 1.0f : x;
    return (int)(x * 32767.0f);             // f32 product, truncation toward zero
}

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
__device__ __forceinline__ unsigned b64c(unsigned v)
{
    // RFC 4648 alphabet without a table: A-Z a-z 0-9 + /
    v &= 63u;
    return v < 26 ? 'A' + v : v < 52 ? 'a' + (v - 26) : v < 62 ? '0' + (v - 52) : v == 62 ? '+' : '/';
}

// one lane = 6 samples = 12 bytes -> 16 characters (one 16-byte store).  The 6-sample groups
// of a workgroup are 1536 consecutive floats: they are fetched with coalesced 16-byte loads
// into LDS and picked up from there (a lane reading its own 24 bytes from global memory
// touches every cache line six times).
constexpr int kB64Tile = kBlock * 6;             // samples per workgroup step

__device__ __forceinline__ void b64_encode16(const float (&x)[6], unsigned char (&ch)[16])
{
    unsigned char by[12];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int q = quant1(x[j]);
        by[2 * j] = (unsigned char)(q & 0xff);
        by[2 * j + 1] = (unsigned char)((q >> 8) & 0xff);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const unsigned v = ((unsigned)by[3 * t] << 16) | ((unsigned)by[3 * t + 1] << 8) | by[3 * t + 2];
        ch[4 * t] = (unsigned char)b64c(v >> 18);
        ch[4 * t + 1] = (unsigned char)b64c(v >> 12);
        ch[4 * t + 2] = (unsigned char)b64c(v >> 6);
        ch[4 * t + 3] = (unsigned char)b64c(v);
    }
}

__global__ __launch_bounds__(kBlock) void pcm16_base64_kernel(const float* __restrict__ in, size_t n,
                                                              char* __restrict__ out, int vec_ok)
{
    __shared__ __attribute__((aligned(16))) float tile[kB64Tile];
    const size_t n_bytes = 2 * n;
    const size_t n_tiles = (n + kB64Tile - 1) / kB64Tile;
    const int tid = threadIdx.x;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t s0 = t * kB64Tile;
        __syncthreads();                                  // the previous tile has been read
        if (vec_ok && s0 + kB64Tile <= n) {
            const float4* src = reinterpret_cast<const float4*>(in + s0);
            for (int i = tid; i < kB64Tile / 4; i += kBlock) reinterpret_cast<float4*>(tile)[i] = src[i];
        } else {
            for (int i = tid; i < kB64Tile; i += kBlock) tile[i] = s0 + i < n ? in[s0 + i] : 0.0f;
        }
        __syncthreads();
        const size_t g = t * kBlock + tid;                // 6-sample group of this lane
        const size_t byte0 = g * 12;
        if (byte0 >= n_bytes) continue;
        float x[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) x[j] = tile[tid * 6 + j];
        unsigned char ch[16];
        b64_encode16(x, ch);
        if (byte0 + 12 <= n_bytes) {
            if (vec_ok) {
                uint4 o;
                o.x = ch[0] | (ch[1] << 8) | (ch[2] << 16) | ((unsigned)ch[3] << 24);
                o.y = ch[4] | (ch[5] << 8) | (ch[6] << 16) | ((unsigned)ch[7] << 24);
                o.z = ch[8] | (ch[9] << 8) | (ch[10] << 16) | ((unsigned)ch[11] << 24);
                o.w = ch[12] | (ch[13] << 8) | (ch[14] << 16) | ((unsigned)ch[15] << 24);
                reinterpret_cast<uint4*>(out)[g] = o;
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c) out[g * 16 + c] = (char)ch[c];
            }
        } else {
            // last, partial group: whole 3-byte triples, then '=' padding
            const size_t rem = n_bytes - byte0;          // 2,4,6,8,10
            const size_t full = rem / 3, tail = rem % 3;
            for (size_t c = 0; c < full * 4; ++c) out[g * 16 + c] = (char)ch[c];
            if (tail) {
                char* o = out + g * 16 + full * 4;
                o[0] = (char)ch[full * 4];
                o[1] = (char)ch[full * 4 + 1];       // low bits already zero (samples beyond are 0)
                o[2] = tail == 2 ? (char)ch[full * 4 + 2] : '=';
                o[3] = '=';
            }
        }
    }
}

// ---- VAD energy ----------------------------------------------------------------
// one wave64 per frame: lanes stride the frame (16-byte loads when the frames are 16-byte
// aligned), wavefront-shuffle tree sum.
template <class F>
__device__ __forceinline__ void frame_sweep(const float* __restrict__ x, size_t len, int lane, bool vec, F&& f)
{
    size_t done = 0;
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const size_t n4 = len / 4;
        for (size_t i = lane; i < n4; i += 64) {
            const float4 v = x4[i];
            f(v.x); f(v.y); f(v.z); f(v.w);
        }
        done = n4 * 4;
    }
    for (size_t i = done + lane; i < len; i += 64) f(x[i]);
}

__global__ void vad_energy_kernel(const float* __restrict__ in, size_t n_frames, size_t frame_len,
                                  float* __restrict__ energy, int vec)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t f = wave; f < n_frames; f += n_waves) {
        float s = 0.0f;
        frame_sweep(in + f * frame_len, frame_len, lane, vec != 0, [&](float v) { s += v * v; });
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) energy[f] = frame_len ? s / (float)frame_len : 0.0f;   // vad.rs:158,167
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
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s += __shfl_xor(s, off, 64);
            m = fmaxf(m, __shfl_xor(m, off, 64));
        }
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
            hipLaunchKernelGGL(to_mono2_kernel, dim3(grid_for(n_pairs)), dim3(kBlock), 0, s, in_dev,
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
    hipLaunchKernelGGL(quantize_kernel, dim3(grid_for((n + 3) / 4)), dim3(kBlock), 0, as_stream(stream),
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
    size_t tiles = (n + kB64Tile - 1) / kB64Tile;
    if (tiles > 2048) tiles = 2048;
    hipLaunchKernelGGL(pcm16_base64_kernel, dim3((int)tiles), dim3(kBlock), 0, as_stream(stream), in_dev,
                       n, out_dev, vec_ok);
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
    size_t g = (n_frames * 64 + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    const int vec = frame_len % 4 == 0 && reinterpret_cast<uintptr_t>(in_dev) % 16 == 0;
    hipLaunchKernelGGL(vad_energy_kernel, dim3((int)g), dim3(kBlock), 0, as_stream(stream), in_dev,
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
    if (g > 4096) g = 4096;
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
