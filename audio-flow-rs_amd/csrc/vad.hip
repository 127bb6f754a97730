// vad.hip -- VoiceActivityDetector behind the C ABI
// (reference: src-tauri/src/modules/audio/vad.rs).
// The per-frame energy reduction (vad.rs:157-168) runs on the device
// (pcm_kernels.hip); the exponential smoothing, dBFS conversion and the
// Silence/Speech/Ending machine (vad.rs:102-151) are a few scalar operations
// per frame and stay on the host, frame after frame, exactly in the
// reference's order -- for ONE detector.  A batch of detectors (one per clip: every VoiceActivityDetector owns
// its state, vad.rs:60-70) is scanned on the device, one lane per clip (af_vad_detect_batch*).
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "af_common.hpp"

struct af_vad {
    af_vad_config cfg{};
    float smoothed_energy = 0.0f;
    size_t silence_frames = 0;
    size_t speech_frames = 0;
    int state = AF_VAD_SILENCE;
    af::DevBuf d_in, d_energy;
    std::vector<float> energies;
};

namespace {

float energy_to_dbfs(float e)
{   // vad.rs:171-176
    if (e <= 0.0f) return -INFINITY;
    return 20.0f * log10f(e);
}

int step(af_vad* v, float energy)
{
    // vad.rs:102-104 (separately rounded f32 products and sum)
    const float old_smoothed = v->smoothed_energy;
    volatile float a = v->cfg.smoothing_factor * energy;
    volatile float one_minus = 1.0f - v->cfg.smoothing_factor;
    volatile float b = one_minus * old_smoothed;
    v->smoothed_energy = a + b;
    // vad.rs:108-118
    const float detection = v->cfg.smoothing_factor > 0.0f ? v->smoothed_energy : energy;
    const bool is_speech = energy_to_dbfs(detection) > v->cfg.threshold_db;
    switch (v->state) {   // vad.rs:121-151
    case AF_VAD_SILENCE:
        if (is_speech) {
            v->speech_frames = 1;
            v->silence_frames = 0;
            v->state = AF_VAD_SPEECH;
        }
        break;
    case AF_VAD_SPEECH:
        if (is_speech) {
            v->speech_frames += 1;
            v->silence_frames = 0;
        } else {
            v->silence_frames += 1;
            if (v->silence_frames >= v->cfg.silence_timeout_frames) {
                v->state = v->speech_frames >= v->cfg.min_speech_frames ? AF_VAD_ENDING
                                                                        : AF_VAD_SILENCE;
                v->speech_frames = 0;
            }
        }
        break;
    default:
        v->state = AF_VAD_SILENCE;
        v->silence_frames = 0;
        break;
    }
    return v->state;
}

// The smallest f32 energy e with  20 * log10f(e) > threshold_db  (vad.rs:115-118 with the host's log10f), found by
// bisection over the bit patterns of the positive floats (log10f is monotone): the device then decides with
// `e >= e_star`, bit-identical to the host path and without a device logarithm.  NaN: no energy qualifies.
float speech_energy_cut(float threshold_db)
{
    auto speech = [&](uint32_t bits) {
        float e;
        std::memcpy(&e, &bits, 4);
        return energy_to_dbfs(e) > threshold_db;
    };
    const uint32_t inf_bits = 0x7f800000u;
    if (!speech(inf_bits)) return NAN;                   // threshold +inf or NaN: nothing is speech
    uint32_t lo = 1, hi = inf_bits;                      // invariant: speech(hi)
    if (speech(lo)) return __builtin_bit_cast(float, lo);
    while (hi - lo > 1) {                                // invariant: !speech(lo)
        const uint32_t mid = lo + (hi - lo) / 2;
        if (speech(mid)) hi = mid; else lo = mid;
    }
    float e;
    std::memcpy(&e, &hi, 4);
    return e;
}

// One lane = one clip's VoiceActivityDetector, frames in order (vad.rs:97-154).  EMA with the reference's three
// separately rounded f32 operations (vad.rs:102-104); counters are 32-bit (a clip has < 2^32 frames).
__global__ __launch_bounds__(64) void vad_scan_kernel(const float* __restrict__ energy, size_t n_clips,
                                                      size_t n_frames, size_t row_stride, float sf, float one_minus,
                                                      float e_star, uint32_t timeout, uint32_t min_speech,
                                                      af_vad_lane* __restrict__ lanes, uint8_t* __restrict__ states,
                                                      size_t state_stride)
{
    const size_t clip = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (clip >= n_clips) return;
    float smoothed = 0.0f;
    uint32_t silence = 0, speech = 0, state = AF_VAD_SILENCE;
    if (lanes) {
        const af_vad_lane l = lanes[clip];
        smoothed = l.smoothed_energy; silence = l.silence_frames; speech = l.speech_frames; state = l.state;
    }
    const float* row = energy + clip * row_stride;
    uint8_t* out = states + clip * state_stride;
    const bool use_smoothed = sf > 0.0f;                 // vad.rs:108-112
    auto step = [&](float e) -> uint32_t {
        smoothed = __fadd_rn(__fmul_rn(sf, e), __fmul_rn(one_minus, smoothed));   // vad.rs:103-104
        const float det = use_smoothed ? smoothed : e;
        const bool is_speech = det >= e_star;            // == energy_to_dbfs(det) > threshold_db (false for NaN)
        if (state == AF_VAD_SILENCE) {                   // vad.rs:121-151
            if (is_speech) { speech = 1; silence = 0; state = AF_VAD_SPEECH; }
        } else if (state == AF_VAD_SPEECH) {
            if (is_speech) { speech += 1; silence = 0; }
            else {
                silence += 1;
                if (silence >= timeout) {
                    state = speech >= min_speech ? AF_VAD_ENDING : AF_VAD_SILENCE;
                    speech = 0;
                }
            }
        } else {
            state = AF_VAD_SILENCE;
            silence = 0;
        }
        return state;
    };
    size_t f = 0;
    const bool vec = (reinterpret_cast<uintptr_t>(row) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 4 == 0);
    if (vec) {
        for (; f + 4 <= n_frames; f += 4) {
            const float4 e = *reinterpret_cast<const float4*>(row + f);
            const uint32_t s0 = step(e.x), s1 = step(e.y), s2 = step(e.z), s3 = step(e.w);
            *reinterpret_cast<uint32_t*>(out + f) = s0 | (s1 << 8) | (s2 << 16) | (s3 << 24);
        }
    }
    for (; f < n_frames; ++f) out[f] = (uint8_t)step(row[f]);
    if (lanes) lanes[clip] = af_vad_lane{smoothed, silence, speech, state};
}

}  // namespace

using namespace af;

extern "C" {

af_status af_vad_detect_batch_device(const af_vad_config* cfg, const float* energy_dev, size_t n_clips,
                                     size_t n_frames, size_t row_stride, af_vad_lane* lanes_dev,
                                     uint8_t* states_dev, size_t state_stride, void* stream)
{
    AF_REQUIRE(cfg, "af_vad_detect_batch_device: null config");
    if (n_clips == 0 || n_frames == 0) return AF_OK;
    AF_REQUIRE(energy_dev && states_dev, "af_vad_detect_batch_device: null buffer");
    AF_REQUIRE(row_stride >= n_frames && state_stride >= n_frames, "af_vad_detect_batch_device: stride < n_frames");
    AF_REQUIRE(n_frames < ((size_t)1 << 32), "af_vad_detect_batch_device: a clip must have < 2^32 frames");
    AF_TRY(ensure_device(-1));
    const float sf = cfg->smoothing_factor;
    volatile float one_minus = 1.0f - sf;                // vad.rs:104, rounded once as on the host
    const auto clamp32 = [](size_t v) { return v > 0xffffffffull ? 0xffffffffu : (uint32_t)v; };
    const unsigned grid = (unsigned)((n_clips + 63) / 64);
    hipLaunchKernelGGL(vad_scan_kernel, dim3(grid), dim3(64), 0, as_stream(stream), energy_dev, n_clips, n_frames,
                       row_stride, sf, (float)one_minus, speech_energy_cut(cfg->threshold_db),
                       clamp32(cfg->silence_timeout_frames), clamp32(cfg->min_speech_frames), lanes_dev,
                       states_dev, state_stride);
    AF_HIP(hipGetLastError());
    return AF_OK;
}

af_status af_vad_detect_batch(const af_vad_config* cfg, const float* in, size_t n_clips, size_t n_frames,
                              size_t frame_len, uint8_t* states_out)
{
    AF_REQUIRE(cfg, "af_vad_detect_batch: null config");
    if (n_clips == 0 || n_frames == 0) return AF_OK;
    AF_REQUIRE(states_out && (in || frame_len == 0), "af_vad_detect_batch: null buffer");
    AF_TRY(ensure_device(-1));
    const size_t total = n_clips * n_frames;
    DevBuf d_in, d_energy, d_states;
    AF_TRY(d_energy.reserve(total * sizeof(float)));
    AF_TRY(d_states.reserve(total));
    if (frame_len > 0) {
        AF_TRY(d_in.reserve(total * frame_len * sizeof(float)));
        AF_HIP(hipMemcpy(d_in.p, in, total * frame_len * sizeof(float), hipMemcpyHostToDevice));
        AF_TRY(af_vad_energy_device(d_in.as<float>(), total, frame_len, d_energy.as<float>(), nullptr));
    } else {
        AF_HIP(hipMemset(d_energy.p, 0, total * sizeof(float)));     // vad.rs:158-160: empty frame -> energy 0
    }
    AF_TRY(af_vad_detect_batch_device(cfg, d_energy.as<float>(), n_clips, n_frames, n_frames, nullptr,
                                      d_states.as<uint8_t>(), n_frames, nullptr));
    AF_HIP(hipMemcpy(states_out, d_states.p, total, hipMemcpyDeviceToHost));
    return AF_OK;
}

void af_vad_default_config(af_vad_config* cfg)
{   // vad.rs:34-43
    if (!cfg) return;
    cfg->threshold_db = -50.0f;
    cfg->smoothing_factor = 0.3f;
    cfg->silence_timeout_frames = 15;
    cfg->min_speech_frames = 3;
}

af_status af_vad_create(const af_vad_config* cfg, af_vad** out)
{
    AF_REQUIRE(out, "af_vad_create: null output");
    *out = nullptr;
    af_vad* v = new (std::nothrow) af_vad();
    if (!v) return fail(AF_ERR_OUT_OF_MEMORY, "out of host memory");
    if (cfg) v->cfg = *cfg; else af_vad_default_config(&v->cfg);
    *out = v;
    return AF_OK;
}

void af_vad_destroy(af_vad* v) { delete v; }

af_status af_vad_detect_frames(af_vad* v, const float* in, size_t n_frames, size_t frame_len,
                               int* states_out)
{
    AF_REQUIRE(v, "af_vad_detect_frames: null handle");
    if (n_frames == 0) return AF_OK;
    AF_REQUIRE(states_out && (in || frame_len == 0), "af_vad_detect_frames: null buffer");
    v->energies.assign(n_frames, 0.0f);
    if (frame_len > 0) {                      // vad.rs:158-160: empty frame -> energy 0
        AF_TRY(ensure_device(-1));
        const size_t n = n_frames * frame_len;
        AF_TRY(v->d_in.reserve(n * sizeof(float)));
        AF_TRY(v->d_energy.reserve(n_frames * sizeof(float)));
        AF_HIP(hipMemcpy(v->d_in.p, in, n * sizeof(float), hipMemcpyHostToDevice));
        AF_TRY(af_vad_energy_device(v->d_in.as<float>(), n_frames, frame_len,
                                    v->d_energy.as<float>(), nullptr));
        AF_HIP(hipMemcpy(v->energies.data(), v->d_energy.p, n_frames * sizeof(float),
                         hipMemcpyDeviceToHost));
    }
    for (size_t f = 0; f < n_frames; ++f) states_out[f] = step(v, v->energies[f]);
    return AF_OK;
}

af_status af_vad_detect(af_vad* v, const float* frame, size_t n, int* state_out)
{
    int st = AF_VAD_SILENCE;
    AF_TRY(af_vad_detect_frames(v, frame, 1, n, &st));
    if (state_out) *state_out = st;
    return AF_OK;
}

void af_vad_reset(af_vad* v)
{   // vad.rs:179-184
    if (!v) return;
    v->smoothed_energy = 0.0f;
    v->silence_frames = 0;
    v->speech_frames = 0;
    v->state = AF_VAD_SILENCE;
}

int af_vad_state(const af_vad* v) { return v ? v->state : AF_VAD_SILENCE; }
float af_vad_energy_db(const af_vad* v) { return v ? energy_to_dbfs(v->smoothed_energy) : -INFINITY; }
size_t af_vad_speech_frame_count(const af_vad* v) { return v ? v->speech_frames : 0; }

}  // extern "C"
