/*
 * af_oracle.h -- CPU oracle for the audio-flow DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker / reported baseline.
 * The product (libaudioflow_dsp.so) never links, loads or calls it.
 *
 * Parity status:
 *   [R] reference-pinned : to_mono, PCM16 quantiser, base64, energy VAD,
 *       ring buffer, 128-frame chunker, equal-rate passthrough.  Each is a
 *       restatement of the cited reference lines and is pinned by the
 *       reference's own unit-test vectors (tests/test_oracle_reference_kat.py).
 *   [B] build-defined    : polyphase resampler, FIR, Hann STFT, mel
 *       filterbank.  The reference has NO implementation and NO tests of
 *       these ("parity unpinned"); the oracle restates docs/OPS.md in float64
 *       and is cross-checked against numpy/scipy in tests/.
 *
 * All citations are relative to /root/reference/src-tauri/src/.
 */
#ifndef AF_ORACLE_H
#define AF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- [R] -- */

/* modules/audio/capture.rs:30-42  AudioFrame::to_mono
 * channels==1 -> copy.  Otherwise `chunks(channels)`: every chunk (the last
 * one may be partial) is summed sequentially in f32 and divided by
 * `channels as f32`.  Returns the number of output samples
 * (= ceil(n/channels)); 0 and no write if channels==0. */
size_t afo_to_mono(const float* in, size_t n, unsigned channels, float* out);

/* modules/network/websocket.rs:246-251  f32 -> i16:
 * (x.clamp(-1,1) * 32767.0) as i16 : f32 multiply, truncate toward zero,
 * saturating, NaN -> 0. */
void afo_quantize_pcm16(const float* in, size_t n, int16_t* out);
/* websocket.rs:249 to_le_bytes + :254 STANDARD.encode: little-endian bytes,
 * RFC 4648 alphabet with '=' padding.  `out` needs 4*ceil(n_bytes/3)+1. */
size_t afo_base64_encode(const uint8_t* in, size_t n_bytes, char* out);
size_t afo_base64_len(size_t n_bytes);

/* modules/audio/vad.rs:157-168  mean-square "energy" (sequential f32). */
float afo_vad_energy(const float* frame, size_t n);
/* vad.rs:171-176  20*log10(e), -inf for e<=0. */
float afo_energy_to_dbfs(float energy);

typedef struct {
    float threshold_db;          /* vad.rs:37 default -50 */
    float smoothing_factor;      /* vad.rs:38 default 0.3 */
    size_t silence_timeout_frames; /* vad.rs:39 default 15 */
    size_t min_speech_frames;    /* vad.rs:40 default 3 */
} afo_vad_config;

enum { AFO_VAD_SILENCE = 0, AFO_VAD_SPEECH = 1, AFO_VAD_ENDING = 2 };

typedef struct {
    afo_vad_config cfg;
    float smoothed_energy;
    size_t silence_frames;
    size_t speech_frames;
    int state;
} afo_vad;

void afo_vad_default_config(afo_vad_config* c);
void afo_vad_init(afo_vad* v, const afo_vad_config* c);     /* vad.rs:80-88 */
int  afo_vad_detect(afo_vad* v, const float* frame, size_t n); /* vad.rs:97-154 */
/* Same state machine driven by a precomputed energy (for checking the GPU
 * energy kernel + host state machine split). */
int  afo_vad_step_energy(afo_vad* v, float energy);
/* n_clips independent detectors over energy[n_clips][n_frames] -> states[n_clips][n_frames] */
void afo_vad_scan_batch(const afo_vad_config* c, const float* energy, size_t n_clips, size_t n_frames,
                        unsigned char* states);
void afo_vad_reset(afo_vad* v);                              /* vad.rs:179-184 */

/* modules/audio/capture.rs:84-161  RingBuffer (capacity-1 usable slots). */
typedef struct {
    float* buf;
    size_t capacity, write_pos, read_pos;
} afo_ring;
int    afo_ring_init(afo_ring* r, size_t capacity);
void   afo_ring_free(afo_ring* r);
size_t afo_ring_write(afo_ring* r, const float* data, size_t n);      /* :103-122 */
/* returns number read; (size_t)-1 encodes the reference's `None` (:131-133) */
size_t afo_ring_read(afo_ring* r, size_t size, float* out);           /* :125-145 */
size_t afo_ring_available(const afo_ring* r);                         /* :148-152 */
void   afo_ring_clear(afo_ring* r);                                   /* :155-160 */

/* modules/audio/resampler.rs:132-166 BatchResampler bookkeeping: how many
 * full 128-frame chunks a process() call hands to the inner resampler and
 * what is carried.  chunk==0 (equal rates) is the reference's non-terminating
 * case (:34-39,:138); the oracle reports it as chunks=0, carry=0
 * (passthrough) -- the documented deviation. */
void afo_chunker_process(size_t carry_in, size_t n_in, size_t chunk,
                         size_t* n_chunks, size_t* carry_out);

/* ---------------------------------------------------------------- [B] -- */

/* docs/OPS.md section "Polyphase resampler".  L/M = (out/g)/(in/g). */
void   afo_resample_ratio(uint32_t in_hz, uint32_t out_hz, uint32_t* L, uint32_t* M);
size_t afo_resample_out_len(size_t n_in, uint32_t L, uint32_t M); /* ceil(n*L/M) */
/* Prototype low-pass, length L*T, periodic Blackman-Harris windowed sinc
 * centred on L*T/2, each phase normalised to unit DC gain.  h[L*T]. */
void   afo_resample_design(uint32_t L, uint32_t M, uint32_t T, double* h);
/* y[m] = sum_{k<T} h[(m*M mod L)+k*L] * x[floor(m*M/L) - k + T/2], zero ext.
 * x0 = absolute index of x[0] in the stream (0 for one-shot), n_total = total
 * stream length for zero extension at the end (n for one-shot); computes
 * outputs m in [m0, m0+n_out). */
void   afo_resample_f64(const float* x, size_t n, const double* h,
                        uint32_t L, uint32_t M, uint32_t T,
                        size_t m0, size_t n_out, double* y);
/* f32-accumulate version (taps rounded to f32, k ascending fmaf-free
 * mul+add); used as the timed CPU baseline. */
void   afo_resample_f32(const float* x, size_t n, const float* h,
                        uint32_t L, uint32_t M, uint32_t T,
                        size_t m0, size_t n_out, float* y);

/* docs/OPS.md "FIR": y[n] = sum_{k<K} h[k] x[n-k], zero initial state. */
void afo_fir_f64(const float* x, size_t n, const float* h, size_t K, double* y);
void afo_fir_f32(const float* x, size_t n, const float* h, size_t K, float* y);

/* docs/OPS.md "STFT".  n_frames = 1 + floor((n - n_fft)/hop), 0 if n<n_fft. */
size_t afo_stft_n_frames(size_t n, size_t n_fft, size_t hop);
void   afo_hann_periodic(size_t N, double* w);   /* 0.5 - 0.5 cos(2 pi n/N) */
/* power[f][k] = |sum_n w[n] x[f*hop+n] e^{-2 pi i k n / n_fft}|^2,
 * k = 0..n_fft/2.  Powers of two: radix-2 FFT in float64; any other n_fft: the DFT sum itself in float64. */
void   afo_stft_power_f64(const float* x, size_t n, size_t n_fft, size_t hop,
                          double* power);
/* direct O(N^2) DFT of one frame -- independent check of the FFT above */
void   afo_dft_power_naive_f64(const float* x, size_t n_fft, const double* w,
                               double* power);

/* docs/OPS.md "Level meter": rms = sqrt(mean x^2), peak = max|x|, dbfs = 20 log10(rms)
 * (float64; specs/0002-design.md:727-732). */
void afo_level_meter_f64(const float* x, size_t n, double* rms, double* peak, double* dbfs);

/* docs/OPS.md "Mel filterbank": HTK mel, triangular, no area normalisation.
 * W[n_mels][n_fft/2+1]. */
double afo_hz_to_mel_htk(double hz);
double afo_mel_to_hz_htk(double mel);
void   afo_mel_weights(double sample_rate, size_t n_fft, size_t n_mels,
                       double f_min, double f_max, double* W);
/* scale 0 = HTK, 1 = Slaney (linear 200/3 Hz per mel below 1 kHz, 27 mels per factor 6.4 above);
 * norm 1 = Slaney area normalisation 2 / (e[m+2] - e[m]) */
void   afo_mel_weights_ex(double sample_rate, size_t n_fft, size_t n_mels, double f_min,
                          double f_max, int scale, int norm, double* W);
void   afo_mel_apply_f64(const double* power, size_t n_frames, size_t n_bins,
                         const double* W, size_t n_mels, double* mel);

/* Whole feature path for one clip: interleaved PCM -> to_mono (f32, [R]) ->
 * STFT power (f64) -> mel (f64).  mel[n_frames][n_mels]. Returns n_frames. */
size_t afo_features_f64(const float* pcm, size_t n_frames_pcm, unsigned channels,
                        double sample_rate, size_t n_fft, size_t hop,
                        size_t n_mels, double f_min, double f_max, double* mel);

/* ------------------------------------------------ timed CPU baseline ---- */
/* float32 port of the same path (iterative radix-2 FFT, f32 twiddles),
 * `n_clips` equal-length clips, OpenMP over clips with `threads` threads
 * (1 = the reference's scalar single-threaded character).  Returns frames
 * per clip. */
size_t afo_features_f32_batch(const float* pcm, size_t n_clips,
                              size_t n_frames_pcm, unsigned channels,
                              double sample_rate, size_t n_fft, size_t hop,
                              size_t n_mels, double f_min, double f_max,
                              float* mel, int threads);
int afo_max_threads(void);

#ifdef __cplusplus
}
#endif
/* [B, unpinned] cubic interpolation mode (docs/OPS.md "Cubic interpolation"; the configuration of resampler.rs:43-49) */
void afo_resample_cubic(const float* x, size_t n, unsigned in_hz, unsigned out_hz, size_t n_out, double* y);

#endif
