// vad.hip -- VoiceActivityDetector behind the C ABI
// (reference: src-tauri/src/modules/audio/vad.rs).
// The per-frame energy reduction (vad.rs:157-168) runs on the device
// (pcm_kernels.hip); the exponential smoothing, dBFS conversion and the
// Silence/Speech/Ending machine (vad.rs:102-151) are a few scalar operations
// per frame and stay on the host, frame after frame, exactly in the
// reference's order.
#include <cmath>
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

}  // namespace

using namespace af;

extern "C" {

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
