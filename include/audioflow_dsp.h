/*
 * audioflow_dsp.h -- C ABI of libaudioflow_dsp.so, the MI355X (gfx950) DSP
 * hot path behind audio-flow's audio module.
 *
 * This is the drop-in boundary: plain `extern "C"`, pointers + sizes, opaque
 * handles, int status codes.  No C++/torch types cross it.  Every entry point
 * names the reference interface (forfd8960/audio-flow-rs, paths relative to
 * src-tauri/src/) it replaces or, for ops the reference only plans
 * (specs/0001-spec.md:266, specs/0002-design.md:1519-1523), the build-defined
 * op in docs/OPS.md -- marked [R] (reference-pinned) or [B] (build-defined).
 *
 * Conventions
 *  - status: 0 = AF_OK, non-zero = error; af_last_error() returns the
 *    thread-local message of the last failing call (the Rust shim maps it to
 *    AudioError::ResamplingFailed(msg), error.rs:109-110).
 *  - nothing aborts or throws across the ABI.
 *  - handles are NOT thread-safe (they mirror `&mut self`,
 *    modules/audio/resampler.rs:71); distinct handles may be used concurrently.
 *  - "*_device" entry points take device pointers and a hipStream_t (as
 *    void*), enqueue work and return without synchronising; the host-buffer
 *    entry points copy in, run, copy out and synchronise.
 *  - there is NO CPU fallback: every compute call needs a gfx950 device and
 *    fails with AF_ERR_NO_DEVICE / AF_ERR_HIP otherwise.
 */
#ifndef AUDIOFLOW_DSP_H
#define AUDIOFLOW_DSP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t af_status;
enum {
    AF_OK = 0,
    AF_ERR_INVALID_ARG = 1,     /* null pointer, zero rate, unsupported shape */
    AF_ERR_BUFFER_TOO_SMALL = 2,
    AF_ERR_NO_DEVICE = 3,       /* no gfx950 device visible */
    AF_ERR_HIP = 4,             /* a HIP runtime call failed (message has it) */
    AF_ERR_OUT_OF_MEMORY = 5,
    AF_ERR_UNSUPPORTED = 6
};

/* sample formats of interleaved PCM handed to the batch entry points */
enum { AF_PCM_F32 = 0, AF_PCM_I16 = 1 };

const char* af_last_error(void);
const char* af_version(void);
/* number of visible HIP devices (0 if none; never fails) */
int         af_device_count(void);
/* Diagnostics only (no reference counterpart): process-wide switches used by the tests and the
 * profiling scripts; value 0 restores the product behaviour.
 *   "features.kernel"     1: the general fused STFT+mel kernel also where the hop-256 kernel applies; 2: the hop-256
 *                         register-FFT kernel (the default there); 3: the hop-256 matrix-core DFT kernel (mel output
 *                         through the banded plan only: AF_ERR_UNSUPPORTED elsewhere; docs/OPS.md "STFT")
 *   "features.wg_per_cu"  workgroups per CU of the fused kernels
 *   "banded.off"          1: resampler / FIR on the thread-per-output kernels instead of MFMA
 *   "chain.unfused"       1: af_chain_* runs resampler and FIR as two launches instead of one
 * The switches are atomics: reading and writing them is race-free, and an entry point reads a switch once per
 * call.  They are process-wide, though: with handles on several threads use the per-handle forms
 * (af_chain_set_option, af_features_set_option) instead, which override them. */
af_status   af_debug_set(const char* key, int value);

/* ------------------------------------------------------------------ [R] */
/* AudioFrame::to_mono, modules/audio/capture.rs:30-42.
 * n_samples interleaved f32 samples, `channels` per frame; a trailing partial
 * frame is summed and still divided by `channels` (Rust `chunks`).
 * channels==1 copies.  Output length = ceil(n_samples/channels). */
size_t    af_to_mono_out_len(size_t n_samples, uint16_t channels);
af_status af_to_mono(const float* in, size_t n_samples, uint16_t channels,
                     float* out, size_t out_cap, size_t* n_out);
af_status af_to_mono_device(const float* in_dev, size_t n_samples, uint16_t channels,
                            float* out_dev, void* stream);

/* WebSocketClient::send_audio / MessageBuilder::audio_message quantiser,
 * modules/network/websocket.rs:246-251 and :340-345:
 * (x.clamp(-1,1) * 32767.0) as i16, little-endian. */
af_status af_quantize_pcm16(const float* in, size_t n, int16_t* out);
af_status af_quantize_pcm16_device(const float* in_dev, size_t n, int16_t* out_dev,
                                   void* stream);
/* websocket.rs:254 STANDARD.encode of the LE i16 bytes: quantise + base64 on
 * the device, `out` receives af_base64_len(2*n) chars (no NUL). */
size_t    af_base64_len(size_t n_bytes);
af_status af_pcm16_base64(const float* in, size_t n, char* out, size_t out_cap,
                          size_t* n_out);
af_status af_pcm16_base64_device(const float* in_dev, size_t n, char* out_dev,
                                 void* stream);

/* VoiceActivityDetector::calculate_energy, modules/audio/vad.rs:157-168,
 * batched: energy[i] = sum(x[i*frame_len .. +frame_len]^2) / frame_len for
 * n_frames consecutive frames (f32, wavefront-shuffle reduction). */
af_status af_vad_energy(const float* in, size_t n_frames, size_t frame_len,
                        float* energy_out);
af_status af_vad_energy_device(const float* in_dev, size_t n_frames, size_t frame_len,
                               float* energy_dev, void* stream);

/* Level meter [B]: AudioLevel{rms, peak, dbfs} of specs/0002-design.md:727-732 (the payload
 * of the `audio_level` event, events/mod.rs:73-75,112-115; spec'd every 50 ms,
 * specs/0001-spec.md:457) for n_windows consecutive windows of window_len samples:
 * rms = sqrt(mean x^2), peak = max |x|, dbfs = 20 log10(rms) (-inf for silence). */
af_status af_level_meter(const float* in, size_t n_windows, size_t window_len,
                         float* rms, float* peak, float* dbfs);
af_status af_level_meter_device(const float* in_dev, size_t n_windows, size_t window_len,
                                float* rms_dev, float* peak_dev, float* dbfs_dev, void* stream);

/* VoiceActivityDetector, vad.rs:21-43 (config), :80 new, :97 detect,
 * :179 reset.  detect() runs the energy reduction on the device and the
 * EMA / dBFS / 3-state machine (vad.rs:102-151) on the host. */
typedef struct af_vad af_vad;
typedef struct {
    float  threshold_db;            /* vad.rs:37, default -50 */
    float  smoothing_factor;        /* vad.rs:38, default 0.3 */
    size_t silence_timeout_frames;  /* vad.rs:39, default 15 */
    size_t min_speech_frames;       /* vad.rs:40, default 3 */
} af_vad_config;
enum { AF_VAD_SILENCE = 0, AF_VAD_SPEECH = 1, AF_VAD_ENDING = 2 };
void      af_vad_default_config(af_vad_config* cfg);
af_status af_vad_create(const af_vad_config* cfg, af_vad** out);
void      af_vad_destroy(af_vad* v);
/* one frame (vad.rs:97); *state_out receives AF_VAD_* */
af_status af_vad_detect(af_vad* v, const float* frame, size_t n, int* state_out);
/* n_frames consecutive frames of frame_len samples; states_out[n_frames] */
af_status af_vad_detect_frames(af_vad* v, const float* in, size_t n_frames,
                               size_t frame_len, int* states_out);
void      af_vad_reset(af_vad* v);                       /* vad.rs:179 */
/* A batch of detectors, one per clip (every VoiceActivityDetector owns its state, vad.rs:60-70), scanned on the
 * device: one lane per clip walks its frames in order (vad.rs:97-154).  The EMA keeps the reference's three
 * separately rounded f32 operations (vad.rs:102-104); the speech decision `20 log10(e) > threshold_db`
 * (vad.rs:115-118) is taken as `e >= e*`, e* being the smallest f32 the host's log10f accepts (found by bisection),
 * so states equal af_vad_detect_frames' exactly.  states: AF_VAD_* as one byte per frame.
 *   energy_dev [n_clips][row_stride]  mean-square energies (af_vad_energy_device), row_stride >= n_frames
 *   lanes_dev  [n_clips] detector state carried in and out (streaming), or NULL = fresh detectors, state dropped
 *   states_dev [n_clips][state_stride] */
typedef struct {
    float    smoothed_energy;   /* vad.rs:63 */
    uint32_t silence_frames;    /* vad.rs:65 */
    uint32_t speech_frames;     /* vad.rs:67 */
    uint32_t state;             /* vad.rs:69, AF_VAD_* */
} af_vad_lane;
af_status af_vad_detect_batch_device(const af_vad_config* cfg, const float* energy_dev, size_t n_clips,
                                     size_t n_frames, size_t row_stride, af_vad_lane* lanes_dev,
                                     uint8_t* states_dev, size_t state_stride, void* stream);
/* host buffers: in [n_clips][n_frames][frame_len] f32 -> states_out [n_clips][n_frames] (energy + scan on the device) */
af_status af_vad_detect_batch(const af_vad_config* cfg, const float* in, size_t n_clips, size_t n_frames,
                              size_t frame_len, uint8_t* states_out);
int       af_vad_state(const af_vad* v);                 /* vad.rs:187 */
float     af_vad_energy_db(const af_vad* v);             /* vad.rs:192 */
size_t    af_vad_speech_frame_count(const af_vad* v);    /* vad.rs:202 */

/* AudioResampler, modules/audio/resampler.rs:23-109.
 * new(input_rate, output_rate) :32; equal rates => exact passthrough
 * (:72-75, pinned by test :185-190) [R].  Unequal rates: the reference wraps
 * rubato's cubic FastFixedIn (never executed by any test; SURVEY section 8
 * A2); here it is the polyphase windowed-sinc resampler of docs/OPS.md [B]
 * (the spec's stated intent, specs/0001-spec.md:266).
 * process() is streaming like the reference's (state carried in the handle):
 * it returns every output whose input support has arrived. */
typedef struct af_resampler af_resampler;
af_status af_resampler_create(uint32_t input_rate, uint32_t output_rate,
                              af_resampler** out);                 /* :32 */
af_status af_resampler_create_48k_to_16k(af_resampler** out);      /* :60 */
void      af_resampler_destroy(af_resampler* r);
uint32_t  af_resampler_input_rate(const af_resampler* r);          /* :96 */
uint32_t  af_resampler_output_rate(const af_resampler* r);         /* :101 */
int       af_resampler_needs_resampling(const af_resampler* r);    /* :106 */
/* upper bound of what the next process(n_in) can return */
size_t    af_resampler_out_len(const af_resampler* r, size_t n_in);
af_status af_resampler_process(af_resampler* r, const float* in, size_t n_in,
                               float* out, size_t out_cap, size_t* n_out); /* :71 */
/* [B] emit the tail (zero extension) so the stream totals ceil(n*L/M) */
af_status af_resampler_flush(af_resampler* r, float* out, size_t out_cap,
                             size_t* n_out);
void      af_resampler_reset(af_resampler* r);

/* One-shot batch resample [B] (BASELINE config 3): n_clips mono clips of
 * n_in samples each, contiguous; out gets n_clips * af_resample_out_len().
 * af_resample_batch (host buffers) keeps its device staging per device between calls and
 * serialises concurrent callers of the SAME device; from page-locked memory it overlaps H2D, kernel
 * and D2H in chunks of whole clips. */
size_t    af_resample_out_len(uint32_t input_rate, uint32_t output_rate, size_t n_in);
af_status af_resample_batch(uint32_t input_rate, uint32_t output_rate,
                            const float* in, size_t n_clips, size_t n_in,
                            float* out, size_t out_cap);
/* Optional cubic mode [B, parity unpinned]: what the reference CONFIGURES but never runs (resampler.rs:43-49: rubato
 * FastFixedIn with PolynomialDegree::Cubic; no reference test has in != out, the crate's source is not in the
 * tree): docs/OPS.md "Cubic interpolation" -- the cubic through x[i-1 .. i+2] at position m * in / out (float64),
 * zero extension, out_len as above.  No anti-alias filter: the polyphase entry points above are the product path. */
af_status af_resample_cubic_batch(uint32_t input_rate, uint32_t output_rate,
                                  const float* in, size_t n_clips, size_t n_in,
                                  float* out, size_t out_cap);
af_status af_resample_cubic_batch_device(uint32_t input_rate, uint32_t output_rate,
                                         const float* in_dev, size_t n_clips, size_t n_in,
                                         float* out_dev, void* stream);
/* af_resample_batch keeps up to 256 MB of device staging per device between calls (what a larger call needed is
 * freed when it returns); this frees the rest.  No reference counterpart (resampler.rs:78 allocates per call). */
void      af_resample_release_staging(void);
af_status af_resample_batch_device(uint32_t input_rate, uint32_t output_rate,
                                   const float* in_dev, size_t n_clips, size_t n_in,
                                   float* out_dev, void* stream);
/* to_mono (capture.rs:30-42) fused into the resampler's loads: `in_dev` holds n_clips clips of
 * n_frames interleaved stereo frames, the result equals af_to_mono_device followed by
 * af_resample_batch_device without the mono intermediate ever reaching HBM (BASELINE config 4's
 * first two stages).  Ratios without the fused path return AF_ERR_UNSUPPORTED;
 * af_resample_has_stereo_path tells beforehand. */
af_status af_resample_batch_stereo_device(uint32_t input_rate, uint32_t output_rate,
                                          const float* in_dev, size_t n_clips, size_t n_frames,
                                          float* out_dev, void* stream);
int       af_resample_has_stereo_path(uint32_t input_rate, uint32_t output_rate);
/* builds the per-ratio tables now, so that later af_resample_batch*_device calls only launch
 * kernels (call it before capturing them into a hipGraph) */
af_status af_resample_prepare(uint32_t input_rate, uint32_t output_rate);
/* diagnostics, host only: the banded-MFMA shape of a ratio.  info[12] = {ok, L, M, block
 * matrices, input samples per block period, MFMA steps per chain, block periods per tile,
 * waves per matrix, waves per workgroup, segment length, segment stride, LDS bytes};
 * returns 0 when the ratio runs on the thread-per-output kernel instead. */
int       af_resample_plan_describe(uint32_t input_rate, uint32_t output_rate, uint32_t channels,
                                    int32_t* info);
/* prototype filter taps h[L*T] as used by the device (f32), for inspection */
af_status af_resample_taps(uint32_t input_rate, uint32_t output_rate,
                           float* taps, size_t cap, uint32_t* L, uint32_t* M, uint32_t* T);

/* BatchResampler, resampler.rs:114-167: 128-frame chunker over an
 * AudioResampler; process() :132 consumes whole chunks and carries the
 * remainder; flush() :150 zero-pads the carried remainder to one chunk.
 * Equal rates: the reference's loop does not terminate (chunk_size 0,
 * :34-39,:138); here it is passthrough (documented deviation). */
typedef struct af_batch_resampler af_batch_resampler;
af_status af_batch_resampler_create(uint32_t input_rate, uint32_t output_rate,
                                    af_batch_resampler** out);     /* :121 */
void      af_batch_resampler_destroy(af_batch_resampler* r);
size_t    af_batch_resampler_out_len(const af_batch_resampler* r, size_t n_in);
size_t    af_batch_resampler_buffered(const af_batch_resampler* r);
af_status af_batch_resampler_process(af_batch_resampler* r, const float* in, size_t n_in,
                                     float* out, size_t out_cap, size_t* n_out); /* :132 */
af_status af_batch_resampler_flush(af_batch_resampler* r, float* out, size_t out_cap,
                                   size_t* n_out);                 /* :150 */

/* ------------------------------------------------------------------ [B] */
/* FirFilter (docs/OPS.md "FIR"): causal direct form, y[n] = sum h[k] x[n-k],
 * same length as the input; the handle carries K-1 samples of history so
 * consecutive process() calls equal one call on the concatenation. */
typedef struct af_fir af_fir;
af_status af_fir_create(const float* taps, size_t n_taps, af_fir** out);
void      af_fir_destroy(af_fir* f);
af_status af_fir_process(af_fir* f, const float* in, size_t n, float* out);
void      af_fir_reset(af_fir* f);
/* stateless batch: n_clips clips of n samples, zero initial state per clip */
af_status af_fir_batch(const float* taps, size_t n_taps, const float* in,
                       size_t n_clips, size_t n, float* out);
af_status af_fir_batch_device(const float* taps_dev, size_t n_taps, const float* in_dev,
                              size_t n_clips, size_t n, float* out_dev, void* stream);
/* the same stateless batch with the taps of a handle (its history is neither read nor
 * written): nothing is allocated or rebuilt, so the call can be captured into a hipGraph */
af_status af_fir_apply_batch_device(const af_fir* f, const float* in_dev, size_t n_clips, size_t n,
                                    float* out_dev, void* stream);

/* FeaturePipeline (docs/OPS.md "STFT", "Mel"): interleaved PCM ->
 * to_mono (capture.rs:30-42 semantics) -> periodic-Hann STFT (n_fft 1024,
 * center=false) -> |X|^2 -> HTK mel filterbank, fused in one kernel; the
 * power spectrum never reaches HBM. */
typedef struct af_features af_features;
typedef struct {
    uint32_t sample_rate;    /* Hz, of the PCM handed in */
    uint32_t channels;       /* 1..16, interleaved; more than 2 (f32 only): af_to_mono_device first, then the mono kernel */
    uint32_t n_fft;          /* the frame length = the DFT size, 32..1024.  1024 (default), 512, 256: register FFT kernels;
                              * any other length (400 = 25 ms at 16 kHz ...): the DFT as a matrix product on the f32
                              * matrix cores (features_dft.hip), same contract, about 10 x the arithmetic */
    uint32_t hop;            /* 64..1024, multiple of 4 */
    uint32_t n_mels;         /* 1..128 */
    float    f_min;          /* Hz */
    float    f_max;          /* Hz; 0 => sample_rate/2 */
    uint32_t sample_format;  /* AF_PCM_F32 or AF_PCM_I16 */
    int32_t  device;         /* HIP device ordinal, -1 = current */
    uint32_t output;         /* AF_OUT_MEL: rows of n_mels values; AF_OUT_POWER: rows of the
                                n_fft/2+1 = 513 values |X[k]|^2 */
    uint32_t log_mel;        /* 1: mel rows hold log10(max(mel, 1e-10)) */
    uint32_t mel_scale;      /* AF_MEL_HTK (default) or AF_MEL_SLANEY */
    uint32_t mel_norm;       /* AF_MEL_NORM_NONE (default) or AF_MEL_NORM_SLANEY (2/(f[m+2]-f[m])) */
    uint32_t channel_mode;   /* AF_CH_DOWNMIX: to_mono first (capture.rs:30-42);
                                AF_CH_SEPARATE (stereo): one spectrogram per channel, rows laid
                                out [clip][channel][frame] */
} af_features_config;
enum { AF_OUT_MEL = 0, AF_OUT_POWER = 1 };
enum { AF_MEL_HTK = 0, AF_MEL_SLANEY = 1 };
enum { AF_MEL_NORM_NONE = 0, AF_MEL_NORM_SLANEY = 1 };
enum { AF_CH_DOWNMIX = 0, AF_CH_SEPARATE = 1 };
void      af_features_default_config(af_features_config* cfg); /* 48 kHz stereo, 1024/256, 80 mels */
af_status af_features_create(const af_features_config* cfg, af_features** out);
void      af_features_destroy(af_features* p);
/* 1 + floor((n_pcm_frames - n_fft)/hop), 0 if shorter than n_fft */
size_t    af_features_n_frames(const af_features* p, size_t n_pcm_frames);
/* floats per output row: n_mels, or 513 with AF_OUT_POWER */
size_t    af_features_row_len(const af_features* p);
/* Describe the batch once (uploads small per-clip tables):
 * clip c occupies PCM frames [clip_offsets[c], clip_offsets[c+1]) of the
 * buffer (n_clips+1 monotone offsets, in frames).  Output rows are the
 * clips' spectrogram frames back to back, [frame][n_mels] f32.
 * *total_frames_out receives the total number of output frames. */
af_status af_features_set_batch(af_features* p, const int64_t* clip_offsets,
                                size_t n_clips, size_t* total_frames_out);
af_status af_features_set_batch_uniform(af_features* p, size_t n_clips,
                                        size_t frames_per_clip,
                                        size_t* total_frames_out);
/* General form: span c = PCM frames [starts[c], starts[c] + lengths[c]) of the buffer.
 * Spans may overlap: a long recording is cut into chunks of C frames (C a multiple of
 * 16*hop) that each carry a right halo of n_fft - hop frames, so chunk k produces exactly
 * the frames that start inside it and the chunked result equals the one-shot framing bit
 * for bit (BASELINE config 5; resampler.rs:132-166 is the reference's carry/flush analogue). */
af_status af_features_set_batch_spans(af_features* p, const int64_t* starts,
                                      const int64_t* lengths, size_t n_spans,
                                      size_t* total_frames_out);
/* per-handle diagnostics (unlike af_debug_set they touch no other handle); value -1: follow af_debug_set
 *   "kernel"     1: the general fused kernel also where the hop-256 kernel applies
 *   "wg_per_cu"  workgroups per CU of the fused kernels
 *   "pool"       0: the hop-256 kernel with every workgroup on its own contiguous range of units only (default: the last
 *                eighth of every range is dealt through a device-wide counter; same rows, bit for bit)
 *   "mel4"       0: the hop-256 kernel's mel phase on 16 x 16 x 4 matrix tiles also where the plan on 16 blocks of
 *                4 x 4 x 1 fits (n_mels a multiple of 4, <= 16 pieces per half of the bands); same rows within the
 *                float32 summation order */
af_status af_features_set_option(af_features* p, const char* key, int value);
/* enqueue the fused kernel on `stream` for the batch set above */
af_status af_features_run_device(af_features* p, const void* pcm_dev, float* mel_dev,
                                 void* stream);
/* host buffers: H2D, run, D2H, synchronise.  pcm_len = elements (samples) the caller's buffer holds;
 * fewer than the batch describes (total PCM frames * channels) is AF_ERR_BUFFER_TOO_SMALL, never an
 * out-of-bounds read (the same holds for af_stft_power, af_chain_run and af_pipeline_run) */
af_status af_features_run(af_features* p, const void* pcm, size_t pcm_len, float* mel, size_t mel_cap);
/* One-shot form of the above for callers without a handle (the `af_pipeline_run(batch desc,
 * pcm, mel_out)` of SURVEY.md section 8(b)): creates the pipeline described by `cfg`, runs the
 * batch whose clip c occupies PCM frames [clip_offsets[c], clip_offsets[c+1]) of `pcm` (host
 * memory), writes the rows to `out` (host, row length as af_features_row_len) and frees
 * everything again.  *total_rows_out may be queried with out == NULL and out_cap == 0. */
typedef struct {
    af_features_config cfg;
    size_t n_clips;
    const int64_t* clip_offsets;   /* n_clips + 1 monotone frame offsets */
} af_batch_desc;
af_status af_pipeline_run(const af_batch_desc* batch, const void* pcm, size_t pcm_len, float* out,
                          size_t out_cap, size_t* total_rows_out);
/* the filterbank the device uses: W[n_mels][n_fft/2+1] f32 */
af_status af_features_mel_weights(const af_features* p, float* W, size_t cap);
/* name and launch geometry of the fused kernel, for profiling harnesses */
const char* af_features_kernel_name(void);
/* diagnostics, host only: the mel plans of a handle, info[81] = {filter groups, MFMA K-steps,
 * 4-step slots of wave 0..3} of the general kernel, then {waves (0: not in use), groups, K-steps,
 * slots of wave 0..7} of the hop-256 kernel, then that kernel's per-slot group ends [8 waves][8 slots]
 * (0: the group continues; else 0x10000 | filters << 8 | first filter) */
int       af_features_plan_describe(const af_features* p, int32_t* info);
/* diagnostics, host only: which kernel family serves the handle and the shape of its round-5 plans, info[12] =
 * {family (1 general fused kernel, 2 hop-256 register-FFT kernel, 4 matrix-product DFT kernel for n_fft other than
 *  1024 / 512 / 256), mel phase on 4 x 4 x 1 blocks available (0 / 1), its 4-step quads of group 0 and 1, its
 *  block pieces of group 0 and 1, then for family 4: K groups of 16 folded samples, bin tiles, mel tiles, waves per
 *  workgroup, LDS bytes per workgroup; last: the 4 x 4 x 1 plan's LDS read-conflict score (slots shared inside a read group)};
 * returns 1 */
int       af_features_kernel_describe(const af_features* p, int32_t* info);
/* diagnostics: enable/disable in-kernel phase stamps; out[32] (optional) receives the
 * [wave][phase] shader-cycle sums recorded since they were enabled */
af_status af_features_debug_stamps(af_features* p, int enable, unsigned long long* out);

/* Stft (docs/OPS.md "STFT"): the power spectrogram alone, rows [frame][n_fft/2+1] -- the
 * fused kernel with AF_OUT_POWER behind the Stft::new(n_fft, hop) shape of SURVEY.md
 * section 8(b).  BASELINE config 1 (160 000 mono samples -> 622 x 513) is this call. */
typedef struct af_stft af_stft;
typedef struct {
    uint32_t channels;       /* 1 or 2 (interleaved; 2 is downmixed first) */
    uint32_t n_fft;          /* 32..1024: the frame length = the DFT size (1024 default; see af_features_config) */
    uint32_t hop;
    uint32_t sample_format;
    int32_t  device;
} af_stft_config;
void      af_stft_default_config(af_stft_config* cfg);         /* mono f32, 1024/256 */
af_status af_stft_create(const af_stft_config* cfg, af_stft** out);
void      af_stft_destroy(af_stft* s);
size_t    af_stft_n_frames(const af_stft* s, size_t n_pcm_frames);
size_t    af_stft_n_bins(const af_stft* s);                    /* n_fft/2 + 1 */
af_status af_stft_set_batch_uniform(af_stft* s, size_t n_clips, size_t frames_per_clip,
                                    size_t* total_frames_out);
af_status af_stft_set_batch(af_stft* s, const int64_t* clip_offsets, size_t n_clips,
                            size_t* total_frames_out);
af_status af_stft_power_device(af_stft* s, const void* pcm_dev, float* power_dev, void* stream);
af_status af_stft_power(af_stft* s, const void* pcm, size_t pcm_len, float* power, size_t power_cap);

/* MelFilterbank (docs/OPS.md "Mel filterbank"): weights and the stand-alone contraction
 * mel[f][m] = sum_k W[m][k] power[f][k] on MFMA, for callers that keep the power spectrum. */
typedef struct af_mel af_mel;
typedef struct {
    uint32_t sample_rate;
    uint32_t n_fft;          /* 32..1024: the frame length = the DFT size (1024 default; see af_features_config) */
    uint32_t n_mels;
    float    f_min, f_max;   /* f_max 0 => sample_rate/2 */
    uint32_t mel_scale, mel_norm, log_mel;
    int32_t  device;
} af_mel_config;
void      af_mel_default_config(af_mel_config* cfg);           /* 16 kHz, 1024, 80 HTK mels */
af_status af_mel_create(const af_mel_config* cfg, af_mel** out);
void      af_mel_destroy(af_mel* m);
af_status af_mel_weights(const af_mel* m, float* W, size_t cap);   /* [n_mels][n_fft/2+1] */
af_status af_mel_apply_device(af_mel* m, const float* power_dev, size_t n_frames,
                              float* mel_dev, void* stream);
af_status af_mel_apply(af_mel* m, const float* power, size_t n_frames, float* mel, size_t mel_cap);

/* The whole chain of BASELINE config 4 behind one handle:
 * interleaved PCM -> to_mono (capture.rs:30-42) -> polyphase resample -> FIR -> STFT -> mel,
 * every stage a kernel of this library enqueued on the caller's stream.  Batches are equal-length
 * clips (set_batch_uniform) or ragged ones (set_batch: AudioFrame is variable-length,
 * capture.rs:13-28); PCM is f32 or the wire format's i16 (websocket.rs:246-251, scaled by 1/32768).
 * Intermediates are allocated by set_batch*, so run_device only launches (hipGraph-capturable).
 * input_rate == output_rate skips the resampler, n_fir_taps == 0 skips the FIR.
 * Where af_chain_fused_plan_describe() says so, downmix + resampler + FIR are ONE launch (the
 * resampled signal before the FIR stays in LDS) and the chain is two launches, for every batch
 * shape and PCM format; the values are those of the separate launches bit for bit when both use
 * the same MFMA step layout (FIRs of 34..65 and 98..129 taps), and within the tolerance of
 * docs/OPS.md otherwise.  Without the fused launch a ragged batch runs as ONE zero-separated strip: a gather,
 * one resampler launch, one FIR launch, the feature kernel -- every clip's rows are those of the clip alone. */
typedef struct af_chain af_chain;
typedef struct {
    uint32_t input_rate;     /* Hz of the PCM handed in */
    uint32_t channels;       /* 1 or 2, interleaved */
    uint32_t output_rate;    /* Hz the spectrogram is computed at */
    const float* fir_taps;   /* host pointer, copied at create; may be NULL */
    size_t   n_fir_taps;     /* 0..1024 */
    uint32_t hop;
    uint32_t n_mels;
    float    f_min, f_max;   /* f_max 0 => output_rate/2 */
    int32_t  device;
    uint32_t sample_format;  /* AF_PCM_F32 | AF_PCM_I16 */
} af_chain_config;
void      af_chain_default_config(af_chain_config* cfg);  /* 48 kHz stereo f32 -> 16 kHz, no FIR, 1024/256, 80 mels */
af_status af_chain_create(const af_chain_config* cfg, af_chain** out);
void      af_chain_destroy(af_chain* c);
size_t    af_chain_n_frames(const af_chain* c, size_t n_pcm_frames);
af_status af_chain_set_batch_uniform(af_chain* c, size_t n_clips, size_t frames_per_clip,
                                     size_t* total_frames_out);
/* ragged batch: clip i = PCM frames [clip_offsets[i], clip_offsets[i+1]) of the buffer (n_clips + 1 monotone
 * offsets); output rows are the clips' spectrogram frames back to back */
af_status af_chain_set_batch(af_chain* c, const int64_t* clip_offsets, size_t n_clips, size_t* total_frames_out);
af_status af_chain_run_device(af_chain* c, const void* pcm_dev, float* mel_dev, void* stream);
/* host buffers; pcm_len counts samples (frames x channels) */
af_status af_chain_run(af_chain* c, const void* pcm, size_t pcm_len, float* mel, size_t mel_cap);
/* per-handle diagnostics (unlike af_debug_set they touch no other handle):
 *   "unfused"  1: resampler and FIR as separate launches; 0: fused where covered; -1: follow af_debug_set
 *   "n_fft"    the frame length of the STFT / mel stage, 32..1024 (the config has no field for it: 1024); e.g. 400 with hop
 *              160 for 25 ms / 10 ms speech frames behind a 48 -> 16 kHz chain.  Forgets the batch: af_chain_set_batch*
 *              must follow.  (af_chain_stream_* keeps 1024.)
 *   "kernel", "wg_per_cu", "mel4": handed to the chain's STFT / mel stage (af_features_set_option); without them that stage
 *              follows af_debug_set("features.kernel") like any feature handle
 *   "pipeline" k > 0: equal-length batches run the two launches of the fused form side by side on two internal
 *              streams, in k chunks of clips (same rows; measured slower than one after the other on MI355X,
 *              DESIGN.md section 9 -- hence 0, off, by default) */
af_status af_chain_set_option(af_chain* c, const char* key, int value);
/* kernels af_chain_run_device enqueues per call for the configuration and batch described (2 when the
 * fused resampler + FIR launch applies; e.g. 3 for config 4 without it: resampler, FIR, STFT / mel) */
int       af_chain_n_launches(const af_chain* c);
/* diagnostics, host only: the shape of the fused resampler + FIR launch for a configuration.
 * info[10] = {ok, resampler MFMA steps, FIR MFMA steps, resampled samples recomputed per tile,
 * padding in front of the FIR history, FIR block periods per column and tile, FIR segment
 * length, FIR segment stride, FIR outputs per tile, LDS bytes}; returns 0 when the chain runs the
 * resampler and the FIR as separate launches. */
int       af_chain_fused_plan_describe(uint32_t input_rate, uint32_t output_rate, uint32_t channels,
                                       size_t n_fir_taps, int32_t* info);

/* ------------------------------------------------------------------ [R] + [B] */
/* WireEncoder: the reference's intended capture -> wire path on a batch of clips, in ONE launch:
 *   interleaved PCM -> AudioFrame::to_mono (modules/audio/capture.rs:30-42) -> AudioResampler::create_48k_to_16k /
 *   process (modules/audio/resampler.rs:60-93; the polyphase resampler of docs/OPS.md) -> the quantiser of
 *   WebSocketClient::send_audio, (x.clamp(-1, 1) * 32767.0) as i16, little-endian (modules/network/websocket.rs:246-251)
 *   -> STANDARD.encode (websocket.rs:254), as wired in specs/0002-design.md:1102-1125.
 * frame_samples == 0: every clip becomes ONE padded base64 string of af_base64_len(2 * out_samples) characters.
 * frame_samples == N (even; 320 = the 20 ms send batch of specs/0002-design.md:1543-1546): every N resampled samples
 * become one padded string of af_wire_message_len() characters, the messages of a clip back to back; a trailing batch of
 * fewer than N samples is not sent (`if self.batch_buffer.len() < 320 { return None; }`).
 * The fused launch covers 1- or 2-channel f32 / i16 PCM, every rate pair whose ratio has a single block matrix (48 k /
 * 32 k -> 16 k, 48 -> 24 k ...) and frame_samples 0 or an even divisor of 3840; the 16 kHz f32 signal and the i16 samples
 * then exist in LDS only.  Other f32 configurations with 1 or 2 channels (44.1 k input ...) run the same stages as two
 * launches on equal-length batches; the text is the same bytes either way ("unfused" option: tests compare them).
 * pcm16 (optional, may be NULL): receives the quantised samples beside the text, clip rows back to back.
 * energy, peak (optional, messages only): per message the mean-square energy of VoiceActivityDetector::calculate_energy
 * (modules/audio/vad.rs:157-168: f32 left fold over the message's resampled samples, bit for bit -- feed it to
 * af_vad_detect_batch*) and max |x| (the `peak` of the audio_level event, events/mod.rs:73-75,112-115), clips' messages back to
 * back (af_wire_total_messages() entries).  In the fused launch the 16 kHz f32 signal exists nowhere else: this is where a VAD
 * gets to see it.  Needs frame_samples to be a multiple of 8. */
typedef struct af_wire af_wire;
typedef struct {
    uint32_t input_rate;      /* 48000 */
    uint32_t output_rate;     /* 16000: "pcm_16000", websocket.rs:274 */
    uint32_t channels;        /* 1 or 2, interleaved */
    uint32_t sample_format;   /* AF_PCM_F32 | AF_PCM_I16 */
    uint32_t frame_samples;   /* 0: one string per clip; N: one string per N resampled samples */
    int32_t  device;
} af_wire_config;
void      af_wire_default_config(af_wire_config* cfg);   /* 48 kHz stereo f32 -> 16 kHz, one string per clip */
af_status af_wire_create(const af_wire_config* cfg, af_wire** out);
void      af_wire_destroy(af_wire* w);
size_t    af_wire_out_samples(const af_wire* w, size_t n_frames);   /* resampled samples of a clip of n_frames */
size_t    af_wire_text_len(const af_wire* w, size_t n_frames);      /* characters of that clip's text */
size_t    af_wire_message_len(const af_wire* w);                    /* characters per message; 0 with frame_samples == 0 */
size_t    af_wire_n_messages(const af_wire* w, size_t n_frames);    /* messages of that clip; 0 with frame_samples == 0 */
af_status af_wire_set_batch_uniform(af_wire* w, size_t n_clips, size_t frames_per_clip, size_t* total_text,
                                    size_t* total_samples);
/* ragged batch: clip i = PCM frames [clip_offsets[i], clip_offsets[i+1]); texts (and PCM16 rows) back to back */
af_status af_wire_set_batch(af_wire* w, const int64_t* clip_offsets, size_t n_clips, size_t* total_text,
                            size_t* total_samples);
af_status af_wire_text_offsets(const af_wire* w, int64_t* text_off /* n_clips + 1 */, size_t cap);
size_t    af_wire_total_messages(const af_wire* w);                 /* messages of the batch described; 0 with frame_samples == 0 */
af_status af_wire_run_device(af_wire* w, const void* pcm_dev, char* text_dev, int16_t* pcm16_dev, float* energy_dev,
                             float* peak_dev, void* stream);
/* host buffers; pcm_len counts samples (frames x channels); energy / peak hold level_cap messages each */
af_status af_wire_run(af_wire* w, const void* pcm, size_t pcm_len, char* text, size_t text_cap, int16_t* pcm16,
                      size_t pcm16_cap, float* energy, float* peak, size_t level_cap);
/* "unfused" 1: the stages as separate launches (diagnostics; equal-length f32 batches) */
af_status af_wire_set_option(af_wire* w, const char* key, int value);
int       af_wire_n_launches(const af_wire* w);                     /* 1 with the fused launch, else 2 */

/* ChainStream: the same chain over an unbounded stream (BASELINE config 5 on the full chain).  The handle owns
 * page-locked staging, the input history the resampler, the FIR (K - 1 samples) and the STFT (n_fft - hop
 * samples) need, and ONE captured hipGraph per step: H2D of the new frames, the fused resampler + FIR launch,
 * the STFT / mel launch, the history move, D2H of the rows.  A step is a whole number of tiles of both kernels
 * (af_chain_stream_step_frames() input frames, the multiple of the step quantum nearest to the request) and the
 * tiles are anchored at absolute sample 0, so the rows of all pushes and the flush, in order, equal af_chain_run
 * on the whole stream bit for bit.  Needs a configuration with the fused launch (af_chain_fused_plan_describe);
 * others get AF_ERR_UNSUPPORTED.  Reference analogue of the carry / flush: BatchResampler,
 * modules/audio/resampler.rs:132-166; of the intended pipeline: specs/0002-design.md:1102-1125. */
typedef struct af_chain_stream af_chain_stream;
af_status af_chain_stream_create(const af_chain_config* cfg, size_t step_frames, af_chain_stream** out);
void      af_chain_stream_destroy(af_chain_stream* s);
size_t    af_chain_stream_step_frames(const af_chain_stream* s);        /* input frames per replayed step */
size_t    af_chain_stream_rows_per_step(const af_chain_stream* s);
size_t    af_chain_stream_row_len(const af_chain_stream* s);
size_t    af_chain_stream_total_rows(const af_chain_stream* s, uint64_t n_frames);  /* after push(es) + flush */
size_t    af_chain_stream_max_rows(const af_chain_stream* s, size_t n_frames);      /* of the next push of n_frames */
af_status af_chain_stream_push(af_chain_stream* s, const void* pcm, size_t n_frames, float* rows_out,
                               size_t rows_cap, size_t* n_rows);
/* rows the next af_chain_stream_flush() hands out (total_rows(frames pushed) minus the rows returned so far): can exceed
 * rows_per_step by one when the resampler's look-ahead spans a hop -- size the flush buffer with this, not with
 * rows_per_step (reference analogue: the final partial chunk of BatchResampler::flush, resampler.rs:150-166) */
size_t    af_chain_stream_flush_rows(const af_chain_stream* s);
af_status af_chain_stream_flush(af_chain_stream* s, float* rows_out, size_t rows_cap, size_t* n_rows);
af_status af_chain_stream_reset(af_chain_stream* s);
/* Time sharding of ONE long recording across GPUs (BASELINE config 5): every rank owns a
 * contiguous range of steps.  af_chain_stream_seek(s, k, &first) continues as if k steps had already run: the frames
 * pushed next must be the recording's frames from `first` on -- the step's history (af_chain_stream_history_frames(),
 * no rows) and then the steps' own frames; step 0 consumes af_chain_stream_first_step_frames() frames, every later
 * step af_chain_stream_step_frames().  Rows come out from row k * rows_per_step on and equal the unbroken stream's
 * rows bit for bit (a steady step's launch does not depend on its index); the rank that holds the end of the
 * recording flushes.  No collective: every rank reads its history from the source like the rest of its range. */
size_t    af_chain_stream_first_step_frames(const af_chain_stream* s);
size_t    af_chain_stream_history_frames(const af_chain_stream* s);
af_status af_chain_stream_seek(af_chain_stream* s, uint64_t step, uint64_t* first_frame);
/* stats[4] = {device ms of the last replayed step, host wall ms of it, steps so far, rows so far} */
af_status af_chain_stream_stats(const af_chain_stream* s, double* stats);

/* ------------------------------------------------------------------ [R]
 * RingBuffer, modules/audio/capture.rs:84-161 (tests :426-561): mutex-guarded f32 ring between
 * the capture callback and the consumer.  capacity - 1 samples are usable (:111); a write that
 * does not fit is truncated and returns what it stored (:103-122); a read returns
 * min(size, available) samples; *is_none (optional) = 1 is the reference's `None` for an empty
 * ring, as opposed to `Some(vec![])` for size == 0 (:125-145). */
typedef struct af_ring af_ring;
af_status af_ring_create(size_t capacity_samples, af_ring** out);       /* RingBuffer::new */
void      af_ring_destroy(af_ring* r);
size_t    af_ring_capacity(const af_ring* r);
size_t    af_ring_write(af_ring* r, const float* data, size_t n);       /* samples stored */
af_status af_ring_read(af_ring* r, size_t size, float* out, size_t out_cap, size_t* n_read, int* is_none);
size_t    af_ring_available(af_ring* r);
void      af_ring_clear(af_ring* r);

/* ------------------------------------------------------------------ [B]
 * FeatureStream: the fused STFT+mel path over an unbounded stream in fixed steps (BASELINE
 * config 5).  The handle owns page-locked staging, the device buffer [n_fft - hop carried frames |
 * step_frames new frames], the carry and ONE hipGraph (H2D copy, fused kernel, carry, D2H copy)
 * captured at create and replayed per step; step k emits the step_frames / hop frames that start in
 * [k step_frames, (k + 1) step_frames) once their last sample has arrived.  The rows of all pushes +
 * the flush, in order, are bit-identical to one-shot framing of the whole stream (frame f = samples
 * [f hop, f hop + n_fft)).  Reference analogue of the carry / flush: BatchResampler,
 * modules/audio/resampler.rs:132-166; of the feeder loop: capture.rs:247-325. */
typedef struct af_stream af_stream;
typedef struct {
    af_features_config features;   /* channel_mode must be AF_CH_DOWNMIX */
    size_t step_frames;            /* PCM frames per replayed step: a multiple of 16 * hop, >= n_fft */
} af_stream_config;
void      af_stream_default_config(af_stream_config* cfg);              /* features defaults, 2^20 frames */
af_status af_stream_create(const af_stream_config* cfg, af_stream** out);
void      af_stream_destroy(af_stream* s);
size_t    af_stream_rows_per_step(const af_stream* s);                  /* step_frames / hop */
size_t    af_stream_row_len(const af_stream* s);
size_t    af_stream_total_rows(const af_stream* s, uint64_t n_frames);  /* after push(es) + flush */
size_t    af_stream_max_rows(const af_stream* s, size_t n_frames);      /* of the next push of n_frames */
/* feed n_frames of interleaved host PCM; every completed step replays the graph and appends its
 * rows to rows_out (capacity in floats); *n_rows = rows written by this call */
af_status af_stream_push(af_stream* s, const void* pcm, size_t n_frames, float* rows_out, size_t rows_cap,
                         size_t* n_rows);
/* end of stream: the rows of the last, partial step */
af_status af_stream_flush(af_stream* s, float* rows_out, size_t rows_cap, size_t* n_rows);
af_status af_stream_reset(af_stream* s);                                /* start a new stream */
/* stats[4] = {device ms of the last replayed step, host wall ms of it, steps so far, rows so far} */
af_status af_stream_stats(const af_stream* s, double* stats);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOFLOW_DSP_H */
