// chain.hip -- the full feature chain of BASELINE config 4 as one handle:
//   interleaved PCM -> to_mono [R capture.rs:30-42] -> polyphase resample [B]
//   -> FIR [B] -> Hann STFT -> |X|^2 -> mel [B]
// Two launches where the shapes allow it (banded::FusedPlan: downmix + resampler + FIR in one kernel,
// the 16 kHz signal before the FIR only exists in LDS; then the fused STFT / mel kernel), otherwise
// every stage is the corresponding device entry point of this library, enqueued back to back on the
// caller's stream.  All intermediate buffers are allocated when the batch is described, so
// af_chain_run_device() only launches kernels (it can be captured into a hipGraph and replayed per
// streaming step).
#include <new>
#include <numeric>
#include <vector>

#include "af_common.hpp"
#include "banded.hpp"

using namespace af;

struct af_chain {
    af_chain_config cfg{};
    af_features* feat = nullptr;
    af_fir* fir = nullptr;           // taps + MFMA fragments, built once
    banded::FusedPlan fused;         // ok: resampler + FIR are one launch
    const float* afrag_res = nullptr;
    DevBuf d_afrag_fir;
    DevBuf d_mono, d_res, d_fir, d_pcm, d_mel;
    size_t n_clips = 0, n_in = 0, n_res = 0, total_frames = 0;
};

namespace {

// Resampler and FIR as one launch when the shapes are covered (banded.hpp, FusedPlan).
af_status plan_fused(af_chain* c, const float* taps)
{
    banded::Plan rp;
    AF_TRY(resampler_fragments(c->cfg.input_rate, c->cfg.output_rate, (int)c->cfg.channels, &rp, &c->afrag_res));
    c->fused = banded::make_fused_plan(rp, (int)c->cfg.n_fir_taps);
    if (!c->fused.ok) return AF_OK;
    DevBuf d_taps;
    AF_TRY(d_taps.reserve(c->cfg.n_fir_taps * sizeof(float)));
    AF_TRY(c->d_afrag_fir.reserve(c->fused.afrag_b_floats() * sizeof(float)));
    AF_HIP(hipMemcpy(d_taps.p, taps, c->cfg.n_fir_taps * sizeof(float), hipMemcpyHostToDevice));
    AF_TRY(banded::build_fused_fir_afrag(c->fused, d_taps.as<float>(), c->d_afrag_fir.as<float>(), nullptr));
    AF_HIP(hipStreamSynchronize(nullptr));
    return AF_OK;
}

}  // namespace

extern "C" {

void af_chain_default_config(af_chain_config* c)
{
    if (!c) return;
    c->input_rate = 48000;
    c->channels = 2;
    c->output_rate = 16000;
    c->fir_taps = nullptr;
    c->n_fir_taps = 0;
    c->hop = 256;
    c->n_mels = 80;
    c->f_min = 0.0f;
    c->f_max = 0.0f;
    c->device = -1;
}

af_status af_chain_create(const af_chain_config* cfg, af_chain** out)
{
    AF_REQUIRE(cfg && out, "af_chain_create: null argument");
    *out = nullptr;
    AF_REQUIRE(cfg->input_rate > 0 && cfg->output_rate > 0, "sample rates must be positive");
    AF_REQUIRE(cfg->channels == 1 || cfg->channels == 2, "channels must be 1 or 2 (got %u)", cfg->channels);
    AF_REQUIRE(cfg->n_fir_taps == 0 || cfg->fir_taps, "fir_taps is null");
    AF_REQUIRE(cfg->n_fir_taps <= 1024, "FIR needs at most 1024 taps (got %zu)", cfg->n_fir_taps);
    AF_TRY(ensure_device(cfg->device));
    af_chain* c = new (std::nothrow) af_chain();
    if (!c) return fail(AF_ERR_OUT_OF_MEMORY, "out of host memory");
    c->cfg = *cfg;
    c->cfg.fir_taps = nullptr;            // the taps are copied to the device below
    af_features_config fc;
    af_features_default_config(&fc);
    fc.sample_rate = cfg->output_rate;
    fc.channels = 1;
    fc.hop = cfg->hop;
    fc.n_mels = cfg->n_mels;
    fc.f_min = cfg->f_min;
    fc.f_max = cfg->f_max;
    fc.device = -1;
    af_status st = af_features_create(&fc, &c->feat);
    // everything run_device needs is built here, so that it only launches kernels
    if (st == AF_OK && cfg->n_fir_taps) st = af_fir_create(cfg->fir_taps, cfg->n_fir_taps, &c->fir);
    if (st == AF_OK) st = af_resample_prepare(cfg->input_rate, cfg->output_rate);
    if (st == AF_OK && cfg->n_fir_taps && cfg->input_rate != cfg->output_rate) st = plan_fused(c, cfg->fir_taps);
    if (st != AF_OK) { af_chain_destroy(c); return st; }
    *out = c;
    return AF_OK;
}

void af_chain_destroy(af_chain* c)
{
    if (!c) return;
    af_features_destroy(c->feat);
    af_fir_destroy(c->fir);
    delete c;
}

size_t af_chain_n_frames(const af_chain* c, size_t n_pcm_frames)
{
    if (!c) return 0;
    const size_t n_res = af_resample_out_len(c->cfg.input_rate, c->cfg.output_rate, n_pcm_frames);
    return af_features_n_frames(c->feat, n_res);
}

af_status af_chain_set_batch_uniform(af_chain* c, size_t n_clips, size_t frames_per_clip,
                                     size_t* total_frames_out)
{
    AF_REQUIRE(c, "af_chain_set_batch_uniform: null handle");
    c->n_clips = n_clips;
    c->n_in = frames_per_clip;
    c->n_res = af_resample_out_len(c->cfg.input_rate, c->cfg.output_rate, frames_per_clip);
    if (c->cfg.channels == 2) AF_TRY(c->d_mono.reserve(n_clips * frames_per_clip * sizeof(float)));
    if (c->cfg.input_rate != c->cfg.output_rate) AF_TRY(c->d_res.reserve(n_clips * c->n_res * sizeof(float)));
    if (c->cfg.n_fir_taps) AF_TRY(c->d_fir.reserve(n_clips * c->n_res * sizeof(float)));
    size_t total = 0;
    AF_TRY(af_features_set_batch_uniform(c->feat, n_clips, c->n_res, &total));
    c->total_frames = total;
    if (total_frames_out) *total_frames_out = total;
    return AF_OK;
}

af_status af_chain_run_device(af_chain* c, const float* pcm_dev, float* mel_dev, void* stream)
{
    AF_REQUIRE(c, "af_chain_run_device: null handle");
    if (c->n_clips == 0 || c->n_in == 0) return AF_OK;
    AF_REQUIRE(pcm_dev && (mel_dev || c->total_frames == 0), "af_chain_run_device: null buffer");
    const float* cur = pcm_dev;
    const bool resample = c->cfg.input_rate != c->cfg.output_rate;
    if (c->fused.ok && !debug_knob(kKnobChainUnfused) && banded::fits(c->fused.a, (int64_t)c->n_in)) {
        AF_TRY(banded::launch_fused(c->fused, c->afrag_res, c->d_afrag_fir.as<float>(), cur, (int64_t)c->n_in,
                                    (int64_t)c->n_in, c->d_fir.as<float>(), (int64_t)c->n_res, (int64_t)c->n_res,
                                    (int64_t)c->n_clips, as_stream(stream)));
        return af_features_run_device(c->feat, c->d_fir.as<float>(), mel_dev, stream);
    }
    if (c->cfg.channels == 2 && resample &&
        af_resample_has_stereo_path(c->cfg.input_rate, c->cfg.output_rate)) {
        // downmix fused into the resampler's loads: the mono signal never reaches HBM
        AF_TRY(af_resample_batch_stereo_device(c->cfg.input_rate, c->cfg.output_rate, cur, c->n_clips,
                                               c->n_in, c->d_res.as<float>(), stream));
        cur = c->d_res.as<float>();
    } else {
    if (c->cfg.channels == 2) {
        AF_TRY(af_to_mono_device(cur, c->n_clips * c->n_in * 2, 2, c->d_mono.as<float>(), stream));
        cur = c->d_mono.as<float>();
    }
    if (resample) {
        AF_TRY(af_resample_batch_device(c->cfg.input_rate, c->cfg.output_rate, cur, c->n_clips, c->n_in,
                                        c->d_res.as<float>(), stream));
        cur = c->d_res.as<float>();
    }
    }
    if (c->cfg.n_fir_taps) {
        AF_TRY(af_fir_apply_batch_device(c->fir, cur, c->n_clips, c->n_res, c->d_fir.as<float>(), stream));
        cur = c->d_fir.as<float>();
    }
    return af_features_run_device(c->feat, cur, mel_dev, stream);
}

int af_chain_n_launches(const af_chain* c)
{
    if (!c) return 0;
    const bool resample = c->cfg.input_rate != c->cfg.output_rate;
    if (c->fused.ok && !debug_knob(kKnobChainUnfused) && banded::fits(c->fused.a, (int64_t)c->n_in)) return 2;
    int n = 1;                                                  // the STFT / mel kernel
    if (c->cfg.n_fir_taps) {
        const banded::FirParts fp = banded::make_fir_parts((int)c->cfg.n_fir_taps);
        n += fp.ok() ? fp.n : 1;
    }
    if (resample) n += 1;
    if (c->cfg.channels == 2 &&
        !(resample && af_resample_has_stereo_path(c->cfg.input_rate, c->cfg.output_rate)))
        n += 1;                                                 // stand-alone to_mono
    return n;
}

int af_chain_fused_plan_describe(uint32_t input_rate, uint32_t output_rate, uint32_t channels, size_t n_fir_taps,
                                 int32_t* info)
{
    if (!info || !input_rate || !output_rate || (channels != 1 && channels != 2)) return 0;
    for (int i = 0; i < 10; ++i) info[i] = 0;
    if (input_rate == output_rate || n_fir_taps == 0 || n_fir_taps > 1024) return 0;
    const uint32_t g = std::gcd(input_rate, output_rate);
    const banded::Plan rp = banded::make_plan(output_rate / g, input_rate / g, banded::kResamplerTaps,
                                              banded::kResamplerTaps / 2, (int)channels);
    const banded::FusedPlan f = banded::make_fused_plan(rp, (int)n_fir_taps);
    const int32_t v[10] = {f.ok ? 1 : 0, f.a.NK, f.NKB, f.H2, f.pad, f.P2, f.seg_len_b, f.seg_stride_b,
                           (int32_t)f.z_per_tile, (int32_t)f.smem};
    if (f.ok)
        for (int i = 0; i < 10; ++i) info[i] = v[i];
    return f.ok ? 1 : 0;
}

af_status af_chain_run(af_chain* c, const float* pcm, size_t pcm_len, float* mel, size_t mel_cap)
{
    AF_REQUIRE(c, "af_chain_run: null handle");
    if (pcm_len < c->n_clips * c->n_in * c->cfg.channels)
        return fail(AF_ERR_BUFFER_TOO_SMALL, "pcm buffer holds %zu samples, the batch describes %zu", pcm_len,
                    c->n_clips * c->n_in * c->cfg.channels);
    const size_t n_out = c->total_frames * c->cfg.n_mels;
    if (mel_cap < n_out) return fail(AF_ERR_BUFFER_TOO_SMALL, "mel buffer holds %zu floats, need %zu", mel_cap, n_out);
    if (c->n_clips == 0 || c->n_in == 0) return AF_OK;
    AF_REQUIRE(pcm && (mel || n_out == 0), "af_chain_run: null buffer");
    const size_t in_bytes = c->n_clips * c->n_in * c->cfg.channels * sizeof(float);
    AF_TRY(c->d_pcm.reserve(in_bytes));
    AF_TRY(c->d_mel.reserve((n_out ? n_out : 1) * sizeof(float)));
    AF_HIP(hipMemcpy(c->d_pcm.p, pcm, in_bytes, hipMemcpyHostToDevice));
    AF_TRY(af_chain_run_device(c, c->d_pcm.as<float>(), c->d_mel.as<float>(), nullptr));
    if (n_out) AF_HIP(hipMemcpy(mel, c->d_mel.p, n_out * sizeof(float), hipMemcpyDeviceToHost));
    else AF_HIP(hipDeviceSynchronize());
    return AF_OK;
}

}  // extern "C"
