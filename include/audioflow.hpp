// audioflow.hpp -- header-only C++17 mirror of the reference's audio module
// (forfd8960/audio-flow-rs, src-tauri/src/modules/audio/*.rs) over the C ABI of
// audioflow_dsp.h.  The reference is Rust; this image has no Rust toolchain, so
// the host side above the C ABI is written in C++ (the Rust shim a maintainer
// would add is in INTEGRATION.md, as source).  Same names, argument meaning and
// error behaviour as the reference:
//
//   AudioFrame::to_mono                 capture.rs:30
//   WireEncoder (to_mono + 48 -> 16 kHz + PCM16 + base64 in one launch)   capture.rs:30-42, resampler.rs:60-93, websocket.rs:244-254
//   RingBuffer::{new_, write, read, available, clear}                     capture.rs:93-160
//   AudioResampler::{new_, create_48k_to_16k, process, input_rate, ...}   resampler.rs:32-108
//   BatchResampler::{new_, process, flush}                                resampler.rs:121-166
//   VoiceActivityDetector::{detect, reset, state, is_speaking, ...}       vad.rs:80-205
//   AudioError::ResamplingFailed(String)                                  error.rs:109
//
// Every failure of the library surfaces as AudioError (a std::runtime_error
// whose what() is "Resampling failed: <msg>", error.rs:109-110), mirroring
// `map_err(|e| AudioError::ResamplingFailed(e.to_string()))` (resampler.rs:49,:85).
#pragma once

#include <cstdint>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "audioflow_dsp.h"

namespace audioflow {

struct AudioError : std::runtime_error {
    af_status status;
    AudioError(af_status st, const std::string& detail)
        : std::runtime_error("Resampling failed: " + detail), status(st) {}
};

namespace detail {
inline void check(af_status st)
{
    if (st != AF_OK) throw AudioError(st, af_last_error());
}
}  // namespace detail

// capture.rs:13-43
struct AudioFrame {
    std::vector<float> samples;
    uint32_t sample_rate = 0;
    uint16_t channels = 1;
    unsigned __int128 timestamp_ns = 0;

    static AudioFrame new_(std::vector<float> samples, uint32_t sample_rate, uint16_t channels,
                           unsigned __int128 timestamp_ns)
    {
        return AudioFrame{std::move(samples), sample_rate, channels, timestamp_ns};
    }

    AudioFrame to_mono() const
    {   // capture.rs:30-42
        if (channels == 1) return *this;
        std::vector<float> mono(af_to_mono_out_len(samples.size(), channels));
        size_t n = 0;
        detail::check(af_to_mono(samples.data(), samples.size(), channels, mono.data(), mono.size(), &n));
        mono.resize(n);
        return AudioFrame{std::move(mono), sample_rate, 1, timestamp_ns};
    }
};

// capture.rs:84-161: mutex-guarded f32 ring, capacity - 1 usable slots
class RingBuffer {
public:
    static RingBuffer new_(size_t capacity_samples)                              // capture.rs:93
    {
        af_ring* h = nullptr;
        detail::check(af_ring_create(capacity_samples, &h));
        return RingBuffer(h);
    }
    RingBuffer(RingBuffer&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    RingBuffer(const RingBuffer&) = delete;
    RingBuffer& operator=(const RingBuffer&) = delete;
    ~RingBuffer() { af_ring_destroy(h_); }

    size_t write(const std::vector<float>& data) { return af_ring_write(h_, data.data(), data.size()); }   // :103
    size_t write(const float* data, size_t n) { return af_ring_write(h_, data, n); }
    std::optional<std::vector<float>> read(size_t size)                           // :125, nullopt = None
    {
        std::vector<float> out(size < af_ring_capacity(h_) ? size : af_ring_capacity(h_));
        size_t n = 0;
        int none = 0;
        detail::check(af_ring_read(h_, size, out.data(), out.size(), &n, &none));
        if (none) return std::nullopt;
        out.resize(n);
        return out;
    }
    size_t available() const { return af_ring_available(h_); }                    // :148
    void clear() { af_ring_clear(h_); }                                           // :155

private:
    explicit RingBuffer(af_ring* h) : h_(h) {}
    af_ring* h_;
};

// resampler.rs:12-109
class AudioResampler {
public:
    static AudioResampler new_(uint32_t input_rate, uint32_t output_rate)
    {
        af_resampler* h = nullptr;
        detail::check(af_resampler_create(input_rate, output_rate, &h));
        return AudioResampler(h);
    }
    static AudioResampler create_48k_to_16k() { return new_(48000, 16000); }   // resampler.rs:60
    AudioResampler() : AudioResampler(new_(48000, 16000)) {}                    // Default, :169
    AudioResampler(AudioResampler&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    AudioResampler& operator=(AudioResampler&& o) noexcept
    {
        if (this != &o) { af_resampler_destroy(h_); h_ = o.h_; o.h_ = nullptr; }
        return *this;
    }
    AudioResampler(const AudioResampler&) = delete;
    AudioResampler& operator=(const AudioResampler&) = delete;
    ~AudioResampler() { af_resampler_destroy(h_); }

    std::vector<float> process(const std::vector<float>& input)
    {   // resampler.rs:71-93
        std::vector<float> out(af_resampler_out_len(h_, input.size()));
        size_t n = 0;
        detail::check(af_resampler_process(h_, input.data(), input.size(), out.data(), out.size(), &n));
        out.resize(n);
        return out;
    }
    std::vector<float> flush()
    {
        std::vector<float> out(af_resampler_out_len(h_, 0));
        size_t n = 0;
        detail::check(af_resampler_flush(h_, out.data(), out.size(), &n));
        out.resize(n);
        return out;
    }
    uint32_t input_rate() const { return af_resampler_input_rate(h_); }           // :96
    uint32_t output_rate() const { return af_resampler_output_rate(h_); }         // :101
    bool needs_resampling() const { return af_resampler_needs_resampling(h_); }   // :106

private:
    explicit AudioResampler(af_resampler* h) : h_(h) {}
    af_resampler* h_ = nullptr;
};

// resampler.rs:114-167
class BatchResampler {
public:
    static BatchResampler new_(uint32_t input_rate, uint32_t output_rate)
    {
        af_batch_resampler* h = nullptr;
        detail::check(af_batch_resampler_create(input_rate, output_rate, &h));
        return BatchResampler(h);
    }
    BatchResampler(BatchResampler&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    BatchResampler(const BatchResampler&) = delete;
    BatchResampler& operator=(const BatchResampler&) = delete;
    ~BatchResampler() { af_batch_resampler_destroy(h_); }

    std::vector<float> process(const std::vector<float>& input)
    {   // resampler.rs:132-147
        std::vector<float> out(af_batch_resampler_out_len(h_, input.size()));
        size_t n = 0;
        detail::check(af_batch_resampler_process(h_, input.data(), input.size(), out.data(), out.size(), &n));
        out.resize(n);
        return out;
    }
    std::vector<float> flush()
    {   // resampler.rs:150-166
        std::vector<float> out(af_batch_resampler_out_len(h_, 0));
        size_t n = 0;
        detail::check(af_batch_resampler_flush(h_, out.data(), out.size(), &n));
        out.resize(n);
        return out;
    }
    size_t buffered() const { return af_batch_resampler_buffered(h_); }

private:
    explicit BatchResampler(af_batch_resampler* h) : h_(h) {}
    af_batch_resampler* h_ = nullptr;
};

// vad.rs:21-54
struct VadConfig {
    float threshold_db = -50.0f;
    float smoothing_factor = 0.3f;
    size_t silence_timeout_frames = 15;
    size_t min_speech_frames = 3;
};
enum class VadState { Silence = AF_VAD_SILENCE, Speech = AF_VAD_SPEECH, Ending = AF_VAD_ENDING };

// vad.rs:60-205
class VoiceActivityDetector {
public:
    explicit VoiceActivityDetector(const VadConfig& c = VadConfig())
    {
        af_vad_config cfg{c.threshold_db, c.smoothing_factor, c.silence_timeout_frames, c.min_speech_frames};
        detail::check(af_vad_create(&cfg, &h_));
    }
    VoiceActivityDetector(const VoiceActivityDetector&) = delete;
    VoiceActivityDetector& operator=(const VoiceActivityDetector&) = delete;
    ~VoiceActivityDetector() { af_vad_destroy(h_); }

    VadState detect(const std::vector<float>& frame)
    {   // vad.rs:97-154
        int st = 0;
        detail::check(af_vad_detect(h_, frame.data(), frame.size(), &st));
        return static_cast<VadState>(st);
    }
    void reset() { af_vad_reset(h_); }                                            // :179
    // A batch of independent detectors, one per clip (vad.rs:60-70), scanned on the device:
    // clips [n_clips][n_frames * frame_len] -> states [n_clips][n_frames] (VadState values as bytes)
    static std::vector<uint8_t> detect_batch(const VadConfig& c, const std::vector<float>& clips, size_t n_clips,
                                             size_t frame_len)
    {
        af_vad_config cfg{c.threshold_db, c.smoothing_factor, c.silence_timeout_frames, c.min_speech_frames};
        const size_t n_frames = (n_clips && frame_len) ? clips.size() / n_clips / frame_len : 0;
        std::vector<uint8_t> out(n_clips * n_frames);
        detail::check(af_vad_detect_batch(&cfg, clips.data(), n_clips, n_frames, frame_len, out.data()));
        return out;
    }
    VadState state() const { return static_cast<VadState>(af_vad_state(h_)); }    // :187
    float energy_db() const { return af_vad_energy_db(h_); }                      // :192
    bool is_speaking() const { return state() == VadState::Speech; }              // :197
    size_t speech_frame_count() const { return af_vad_speech_frame_count(h_); }   // :202

private:
    af_vad* h_ = nullptr;
};

// websocket.rs:246-251
inline std::vector<int16_t> quantize_pcm16(const std::vector<float>& audio)
{
    std::vector<int16_t> out(audio.size());
    detail::check(af_quantize_pcm16(audio.data(), audio.size(), out.data()));
    return out;
}

// websocket.rs:246-254: the quantiser followed by STANDARD.encode -- the `audio_base_64` payload of send_audio
inline std::string pcm16_base64(const std::vector<float>& audio)
{
    std::string out(af_base64_len(2 * audio.size()), '\0');
    size_t n = 0;
    detail::check(af_pcm16_base64(audio.data(), audio.size(), out.empty() ? nullptr : &out[0], out.size(), &n));
    out.resize(n);
    return out;
}

// specs/0002-design.md:727-732 (payload of the audio_level event, events/mod.rs:73-75)
struct AudioLevel {
    float rms, peak, dbfs;
};
inline std::vector<AudioLevel> level_meter(const std::vector<float>& samples, size_t window_len)
{
    const size_t n = window_len ? samples.size() / window_len : 0;
    std::vector<float> r(n), p(n), d(n);
    detail::check(af_level_meter(samples.data(), n, window_len, r.data(), p.data(), d.data()));
    std::vector<AudioLevel> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = {r[i], p[i], d[i]};
    return out;
}

// build-defined siblings (docs/OPS.md)
class FirFilter {
public:
    static FirFilter new_(const std::vector<float>& taps)
    {
        af_fir* h = nullptr;
        detail::check(af_fir_create(taps.data(), taps.size(), &h));
        return FirFilter(h);
    }
    FirFilter(FirFilter&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    FirFilter(const FirFilter&) = delete;
    FirFilter& operator=(const FirFilter&) = delete;
    ~FirFilter() { af_fir_destroy(h_); }
    std::vector<float> process(const std::vector<float>& input)
    {
        std::vector<float> out(input.size());
        detail::check(af_fir_process(h_, input.data(), input.size(), out.data()));
        return out;
    }
    void reset() { af_fir_reset(h_); }

private:
    explicit FirFilter(af_fir* h) : h_(h) {}
    af_fir* h_ = nullptr;
};

struct MelSpectrogram {
    std::vector<float> data;   // [n_frames][n_mels]
    size_t n_frames = 0, n_mels = 0;
};

// Stft::new(n_fft, hop) [B]: power spectrogram rows [frame][n_fft/2+1]
class Stft {
public:
    // n_fft 32..1024 (1024 default; 1024 / 512 / 256 on the register FFT kernels, any other length as a matrix-product DFT):
    // the Stft::new(n_fft, hop) shape of SURVEY.md section 8 A6
    explicit Stft(uint32_t hop = 256, uint32_t channels = 1, uint32_t n_fft = 1024)
    {
        af_stft_config c;
        af_stft_default_config(&c);
        c.hop = hop;
        c.channels = channels;
        c.n_fft = n_fft;
        detail::check(af_stft_create(&c, &h_));
    }
    Stft(const Stft&) = delete;
    Stft& operator=(const Stft&) = delete;
    ~Stft() { af_stft_destroy(h_); }
    size_t n_frames(size_t n_pcm_frames) const { return af_stft_n_frames(h_, n_pcm_frames); }
    size_t n_bins() const { return af_stft_n_bins(h_); }
    // one clip of interleaved PCM -> [n_frames][n_bins]
    std::vector<float> power(const std::vector<float>& pcm, uint32_t channels = 1)
    {
        size_t total = 0;
        detail::check(af_stft_set_batch_uniform(h_, 1, pcm.size() / channels, &total));
        std::vector<float> out(total * n_bins());
        detail::check(af_stft_power(h_, pcm.data(), pcm.size(), out.data(), out.size()));
        return out;
    }

private:
    af_stft* h_ = nullptr;
};

// MelFilterbank::new(sr, n_fft, n_mels) [B]
class MelFilterbank {
public:
    explicit MelFilterbank(uint32_t sample_rate = 16000, uint32_t n_mels = 80)
    {
        af_mel_default_config(&cfg_);
        cfg_.sample_rate = sample_rate;
        cfg_.n_mels = n_mels;
        detail::check(af_mel_create(&cfg_, &h_));
    }
    explicit MelFilterbank(const af_mel_config& cfg) : cfg_(cfg) { detail::check(af_mel_create(&cfg_, &h_)); }
    MelFilterbank(const MelFilterbank&) = delete;
    MelFilterbank& operator=(const MelFilterbank&) = delete;
    ~MelFilterbank() { af_mel_destroy(h_); }
    std::vector<float> weights() const
    {
        std::vector<float> W((size_t)cfg_.n_mels * (cfg_.n_fft / 2 + 1));
        detail::check(af_mel_weights(h_, W.data(), W.size()));
        return W;
    }
    // power rows [n_frames][n_fft/2+1] -> mel rows [n_frames][n_mels]
    std::vector<float> apply(const std::vector<float>& power)
    {
        const size_t n = power.size() / (cfg_.n_fft / 2 + 1);
        std::vector<float> mel(n * cfg_.n_mels);
        detail::check(af_mel_apply(h_, power.data(), n, mel.data(), mel.size()));
        return mel;
    }

private:
    af_mel_config cfg_;
    af_mel* h_ = nullptr;
};

class FeaturePipeline {
public:
    explicit FeaturePipeline(const af_features_config& cfg) : cfg_(cfg)
    {
        detail::check(af_features_create(&cfg_, &h_));
    }
    FeaturePipeline() : FeaturePipeline(default_config()) {}
    FeaturePipeline(const FeaturePipeline&) = delete;
    FeaturePipeline& operator=(const FeaturePipeline&) = delete;
    ~FeaturePipeline() { af_features_destroy(h_); }

    static af_features_config default_config()
    {
        af_features_config c;
        af_features_default_config(&c);
        return c;
    }
    size_t n_frames(size_t n_pcm_frames) const { return af_features_n_frames(h_, n_pcm_frames); }

    // FeaturePipeline::process_batch(&[AudioFrame]) -> Result<Vec<MelSpectrogram>, AudioError>
    std::vector<MelSpectrogram> process_batch(const std::vector<AudioFrame>& frames)
    {
        std::vector<int64_t> off(frames.size() + 1, 0);
        std::vector<float> pcm;
        for (size_t i = 0; i < frames.size(); ++i) {
            if (frames[i].channels != cfg_.channels || frames[i].sample_rate != cfg_.sample_rate)
                throw AudioError(AF_ERR_INVALID_ARG, "frame format does not match the pipeline");
            const size_t nf = frames[i].samples.size() / cfg_.channels;
            off[i + 1] = off[i] + (int64_t)nf;
            pcm.insert(pcm.end(), frames[i].samples.begin(), frames[i].samples.begin() + nf * cfg_.channels);
        }
        size_t total = 0;
        detail::check(af_features_set_batch(h_, off.data(), frames.size(), &total));
        std::vector<float> mel(total * cfg_.n_mels);
        detail::check(af_features_run(h_, pcm.data(), pcm.size(), mel.data(), mel.size()));
        std::vector<MelSpectrogram> out;
        size_t row = 0;
        for (size_t i = 0; i < frames.size(); ++i) {
            // AF_CH_SEPARATE: channel-0 rows then channel-1 rows of the clip
            const size_t nf = (cfg_.channel_mode == AF_CH_SEPARATE ? 2 : 1) * n_frames((size_t)(off[i + 1] - off[i]));
            MelSpectrogram m;
            m.n_frames = nf;
            m.n_mels = cfg_.n_mels;
            m.data.assign(mel.begin() + row * cfg_.n_mels, mel.begin() + (row + nf) * cfg_.n_mels);
            out.push_back(std::move(m));
            row += nf;
        }
        return out;
    }

private:
    af_features_config cfg_;
    af_features* h_ = nullptr;
};

// BASELINE config 5: the fused path over an unbounded stream in fixed, hipGraph-replayed steps
// (af_stream_*).  push() returns the rows of every step it completed, flush() the last partial step.
class FeatureStream {
public:
    explicit FeatureStream(size_t step_frames = (size_t)1 << 20, const af_features_config* features = nullptr)
    {
        af_stream_config cfg;
        af_stream_default_config(&cfg);
        if (features) cfg.features = *features;
        cfg.step_frames = step_frames;
        channels_ = cfg.features.channels;
        detail::check(af_stream_create(&cfg, &h_));
    }
    FeatureStream(const FeatureStream&) = delete;
    FeatureStream& operator=(const FeatureStream&) = delete;
    ~FeatureStream() { af_stream_destroy(h_); }

    size_t row_len() const { return af_stream_row_len(h_); }
    size_t total_rows(uint64_t n_frames) const { return af_stream_total_rows(h_, n_frames); }
    std::vector<float> push(const std::vector<float>& pcm)
    {
        const size_t n_frames = pcm.size() / channels_;
        std::vector<float> rows(af_stream_max_rows(h_, n_frames) * row_len());
        size_t n = 0;
        detail::check(af_stream_push(h_, pcm.data(), n_frames, rows.data(), rows.size(), &n));
        rows.resize(n * row_len());
        return rows;
    }
    std::vector<float> flush()
    {
        std::vector<float> rows(af_stream_rows_per_step(h_) * row_len());
        size_t n = 0;
        detail::check(af_stream_flush(h_, rows.data(), rows.size(), &n));
        rows.resize(n * row_len());
        return rows;
    }
    void reset() { detail::check(af_stream_reset(h_)); }

private:
    af_stream* h_ = nullptr;
    uint32_t channels_ = 2;
};

// BASELINE config 4: interleaved PCM -> to_mono -> polyphase resample -> FIR -> STFT -> mel on a batch of
// clips (af_chain_*): equal lengths or ragged (AudioFrame is variable-length, capture.rs:13-28), f32 or the wire
// format's i16 (websocket.rs:246-251).  Two kernel launches where the fused resampler + FIR stage applies.
namespace detail {
inline af_chain_config chain_config(uint32_t input_rate, uint32_t channels, uint32_t output_rate,
                                    const std::vector<float>& fir_taps, uint32_t n_mels, uint32_t sample_format)
{
    af_chain_config cfg;
    af_chain_default_config(&cfg);
    cfg.input_rate = input_rate;
    cfg.channels = channels;
    cfg.output_rate = output_rate;
    cfg.fir_taps = fir_taps.empty() ? nullptr : fir_taps.data();
    cfg.n_fir_taps = fir_taps.size();
    cfg.n_mels = n_mels;
    cfg.sample_format = sample_format;
    return cfg;
}
}  // namespace detail

class ChainPipeline {
public:
    ChainPipeline(uint32_t input_rate, uint32_t channels, uint32_t output_rate, const std::vector<float>& fir_taps,
                  uint32_t n_mels = 80, uint32_t sample_format = AF_PCM_F32)
    {
        const af_chain_config cfg = detail::chain_config(input_rate, channels, output_rate, fir_taps, n_mels, sample_format);
        n_mels_ = n_mels;
        detail::check(af_chain_create(&cfg, &h_));
    }
    ChainPipeline(const ChainPipeline&) = delete;
    ChainPipeline& operator=(const ChainPipeline&) = delete;
    ~ChainPipeline() { af_chain_destroy(h_); }

    size_t n_frames(size_t n_pcm_frames) const { return af_chain_n_frames(h_, n_pcm_frames); }
    // rows [clip][frame][n_mels], clips back to back
    std::vector<float> run(const std::vector<float>& pcm, size_t n_clips, size_t frames_per_clip)
    {
        size_t total = 0;
        detail::check(af_chain_set_batch_uniform(h_, n_clips, frames_per_clip, &total));
        std::vector<float> mel(total * n_mels_);
        detail::check(af_chain_run(h_, pcm.data(), pcm.size(), mel.data(), mel.size()));
        return mel;
    }
    // ragged batch: clip i = PCM frames [clip_offsets[i], clip_offsets[i + 1]); pcm = f32 or i16 samples (T)
    template <class T>
    std::vector<float> run_ragged(const std::vector<T>& pcm, const std::vector<int64_t>& clip_offsets)
    {
        size_t total = 0;
        detail::check(af_chain_set_batch(h_, clip_offsets.data(), clip_offsets.empty() ? 0 : clip_offsets.size() - 1, &total));
        std::vector<float> mel(total * n_mels_);
        detail::check(af_chain_run(h_, pcm.data(), pcm.size(), mel.data(), mel.size()));
        return mel;
    }
    void set_option(const char* key, int value) { detail::check(af_chain_set_option(h_, key, value)); }
    int n_launches() const { return af_chain_n_launches(h_); }

private:
    af_chain* h_ = nullptr;
    uint32_t n_mels_ = 80;
};

// BASELINE config 5 on the whole chain (af_chain_stream_*): push() returns the rows of every step it completed (one
// hipGraph replay each), flush() the rest; all rows in order == ChainPipeline::run on the whole stream, bit for bit.
// Carry / flush analogue in the reference: BatchResampler, resampler.rs:132-166.
class ChainStream {
public:
    ChainStream(size_t step_frames, uint32_t input_rate, uint32_t channels, uint32_t output_rate,
                const std::vector<float>& fir_taps, uint32_t n_mels = 80, uint32_t sample_format = AF_PCM_F32)
        : channels_(channels)
    {
        const af_chain_config cfg = detail::chain_config(input_rate, channels, output_rate, fir_taps, n_mels, sample_format);
        detail::check(af_chain_stream_create(&cfg, step_frames, &h_));
    }
    ChainStream(const ChainStream&) = delete;
    ChainStream& operator=(const ChainStream&) = delete;
    ~ChainStream() { af_chain_stream_destroy(h_); }

    size_t step_frames() const { return af_chain_stream_step_frames(h_); }
    size_t row_len() const { return af_chain_stream_row_len(h_); }
    size_t total_rows(uint64_t n_frames) const { return af_chain_stream_total_rows(h_, n_frames); }
    template <class T>
    std::vector<float> push(const std::vector<T>& pcm)
    {
        const size_t n_frames = pcm.size() / channels_;
        std::vector<float> rows(af_chain_stream_max_rows(h_, n_frames) * row_len());
        size_t n = 0;
        detail::check(af_chain_stream_push(h_, pcm.data(), n_frames, rows.data(), rows.size(), &n));
        rows.resize(n * row_len());
        return rows;
    }
    std::vector<float> flush()
    {
        // (the flush can owe one row more than a step: af_chain_stream_flush_rows is the bound)
        std::vector<float> rows(std::max<size_t>(af_chain_stream_flush_rows(h_), 1) * row_len());
        size_t n = 0;
        detail::check(af_chain_stream_flush(h_, rows.data(), rows.size(), &n));
        rows.resize(n * row_len());
        return rows;
    }
    void reset() { detail::check(af_chain_stream_reset(h_)); }
    // time sharding: continue as if `step` steps had run; the next push starts at the returned frame
    size_t first_step_frames() const { return af_chain_stream_first_step_frames(h_); }
    size_t history_frames() const { return af_chain_stream_history_frames(h_); }
    uint64_t seek(uint64_t step)
    {
        uint64_t first = 0;
        detail::check(af_chain_stream_seek(h_, step, &first));
        return first;
    }

private:
    af_chain_stream* h_ = nullptr;
    uint32_t channels_ = 2;
};

// The reference's capture -> wire path on a batch of clips in ONE launch (af_wire_*): interleaved PCM ->
// AudioFrame::to_mono (capture.rs:30-42) -> AudioResampler::create_48k_to_16k / process (resampler.rs:60-93) ->
// (x.clamp(-1, 1) * 32767.0) as i16, little-endian (websocket.rs:246-251) -> STANDARD.encode (websocket.rs:254).
// frame_samples = 0: one padded base64 string per clip; 320: one per 20 ms send batch (specs/0002-design.md:1543-1546),
// a trailing partial batch is not sent.  T = float or int16_t PCM.
class WireEncoder {
public:
    WireEncoder(uint32_t input_rate = 48000, uint32_t output_rate = 16000, uint32_t channels = 2,
                uint32_t sample_format = AF_PCM_F32, uint32_t frame_samples = 0)
    {
        af_wire_config cfg;
        af_wire_default_config(&cfg);
        cfg.input_rate = input_rate;
        cfg.output_rate = output_rate;
        cfg.channels = channels;
        cfg.sample_format = sample_format;
        cfg.frame_samples = frame_samples;
        detail::check(af_wire_create(&cfg, &h_));
    }
    WireEncoder(const WireEncoder&) = delete;
    WireEncoder& operator=(const WireEncoder&) = delete;
    ~WireEncoder() { af_wire_destroy(h_); }

    size_t out_samples(size_t n_frames) const { return af_wire_out_samples(h_, n_frames); }
    size_t text_len(size_t n_frames) const { return af_wire_text_len(h_, n_frames); }
    size_t message_len() const { return af_wire_message_len(h_); }
    size_t n_messages(size_t n_frames) const { return af_wire_n_messages(h_, n_frames); }
    int n_launches() const { return af_wire_n_launches(h_); }
    void set_option(const char* key, int value) { detail::check(af_wire_set_option(h_, key, value)); }
    // messages only: the next run also fills these with, per message, VoiceActivityDetector::calculate_energy of the message
    // (vad.rs:157-168, bit for bit) and max |x| (the `peak` of the audio_level event, events/mod.rs:73-75); nullptr: off
    void want_level(std::vector<float>* energy, std::vector<float>* peak) { energy_ = energy; peak_ = peak; }
    // the batch's text, clips back to back (cut it with text_len / message_len); pcm16 (optional) gets the samples
    template <class T>
    std::string run(const std::vector<T>& pcm, size_t n_clips, size_t frames_per_clip, std::vector<int16_t>* pcm16 = nullptr)
    {
        size_t total = 0, samples = 0;
        detail::check(af_wire_set_batch_uniform(h_, n_clips, frames_per_clip, &total, &samples));
        return run_described(pcm, total, samples, pcm16);
    }
    template <class T>
    std::string run_ragged(const std::vector<T>& pcm, const std::vector<int64_t>& clip_offsets, std::vector<int16_t>* pcm16 = nullptr)
    {
        size_t total = 0, samples = 0;
        detail::check(af_wire_set_batch(h_, clip_offsets.data(), clip_offsets.empty() ? 0 : clip_offsets.size() - 1, &total, &samples));
        return run_described(pcm, total, samples, pcm16);
    }

private:
    template <class T>
    std::string run_described(const std::vector<T>& pcm, size_t total, size_t samples, std::vector<int16_t>* pcm16)
    {
        std::string text(total, '\0');
        if (pcm16) pcm16->assign(samples, 0);
        std::vector<float>* en = energy_;
        std::vector<float>* pk = peak_;
        const size_t nm = (en || pk) ? af_wire_total_messages(h_) : 0;
        if (en) en->assign(nm, 0.0f);
        if (pk) pk->assign(nm, 0.0f);
        detail::check(af_wire_run(h_, pcm.data(), pcm.size(), text.empty() ? nullptr : &text[0], text.size(),
                                  pcm16 ? pcm16->data() : nullptr, pcm16 ? pcm16->size() : 0, en ? en->data() : nullptr,
                                  pk ? pk->data() : nullptr, nm));
        return text;
    }
    af_wire* h_ = nullptr;
    std::vector<float>* energy_ = nullptr;
    std::vector<float>* peak_ = nullptr;
};

}  // namespace audioflow
