/*
 * af_oracle.c -- CPU oracle (checker) for the audio-flow DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- see af_oracle.h.  Compile with
 *   gcc -O2 -ffp-contract=off -fno-fast-math
 * so that every f32 operation in the [R] functions rounds exactly where the
 * reference's Rust rounds (no FMA contraction, no reassociation).
 *
 * Citations are relative to /root/reference/src-tauri/src/.
 */
#include "af_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ====================================================================== */
/* [R] reference-pinned restatements                                      */
/* ====================================================================== */

/* capture.rs:30-42.  `.chunks(channels)` yields a final short chunk when
 * n % channels != 0; its partial sum is still divided by `channels`
 * (capture.rs:37-38).  `iter().sum::<f32>()` is a left fold from the additive
 * identity -0.0: the reference is edition 2024 (src-tauri/Cargo.toml:4), i.e.
 * rustc >= 1.85, whose `impl Sum for f32` folds from -0.0 (std changed the
 * seed from +0.0 in 1.83; recollection, not verifiable offline).  The seed
 * only shows in a frame whose samples are all -0.0: -0.0 here, +0.0 with the
 * older seed (docs/OPS.md "to_mono"). */
size_t afo_to_mono(const float* in, size_t n, unsigned channels, float* out)
{
    if (channels == 0) return 0;
    if (channels == 1) {                       /* capture.rs:31-33 clone */
        memcpy(out, in, n * sizeof(float));
        return n;
    }
    size_t n_out = 0;
    for (size_t i = 0; i < n; i += channels) {
        size_t end = i + channels < n ? i + channels : n;
        volatile float sum = -0.0f;            /* volatile: keep f32 rounding */
        for (size_t j = i; j < end; ++j) sum = sum + in[j];
        out[n_out++] = sum / (float)channels;  /* capture.rs:38 */
    }
    return n_out;
}

/* websocket.rs:248: (x.clamp(-1.0, 1.0) * 32767.0) as i16.
 * Rust `as i16` from f32: truncate toward zero, saturate, NaN -> 0.
 * f32::clamp propagates NaN. */
void afo_quantize_pcm16(const float* in, size_t n, int16_t* out)
{
    for (size_t i = 0; i < n; ++i) {
        float x = in[i];
        int16_t s;
        if (x != x) {
            s = 0;
        } else {
            if (x < -1.0f) x = -1.0f;
            if (x > 1.0f) x = 1.0f;
            volatile float v = x * 32767.0f;   /* f32 product */
            s = (int16_t)(int32_t)truncf(v);   /* |v| <= 32767: no saturation */
        }
        out[i] = s;
    }
}

size_t afo_base64_len(size_t n_bytes) { return 4 * ((n_bytes + 2) / 3); }

/* websocket.rs:254 base64::engine::general_purpose::STANDARD (padded). */
size_t afo_base64_encode(const uint8_t* in, size_t n, char* out)
{
    static const char tbl[] =
        "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    size_t o = 0, i = 0;
    for (; i + 2 < n; i += 3) {
        uint32_t v = ((uint32_t)in[i] << 16) | ((uint32_t)in[i + 1] << 8) | in[i + 2];
        out[o++] = tbl[(v >> 18) & 63];
        out[o++] = tbl[(v >> 12) & 63];
        out[o++] = tbl[(v >> 6) & 63];
        out[o++] = tbl[v & 63];
    }
    if (i + 1 == n) {
        uint32_t v = (uint32_t)in[i] << 16;
        out[o++] = tbl[(v >> 18) & 63];
        out[o++] = tbl[(v >> 12) & 63];
        out[o++] = '=';
        out[o++] = '=';
    } else if (i + 2 == n) {
        uint32_t v = ((uint32_t)in[i] << 16) | ((uint32_t)in[i + 1] << 8);
        out[o++] = tbl[(v >> 18) & 63];
        out[o++] = tbl[(v >> 12) & 63];
        out[o++] = tbl[(v >> 6) & 63];
        out[o++] = '=';
    }
    out[o] = '\0';
    return o;
}

/* vad.rs:157-168: empty -> 0; sum of squares (left fold, f32) / len as f32. */
float afo_vad_energy(const float* frame, size_t n)
{
    if (n == 0) return 0.0f;
    volatile float sum = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        volatile float sq = frame[i] * frame[i];
        sum = sum + sq;
    }
    return sum / (float)n;
}

/* vad.rs:171-176 */
float afo_energy_to_dbfs(float energy)
{
    if (energy <= 0.0f) return -INFINITY;
    return 20.0f * log10f(energy);
}

void afo_vad_default_config(afo_vad_config* c)
{   /* vad.rs:34-43 */
    c->threshold_db = -50.0f;
    c->smoothing_factor = 0.3f;
    c->silence_timeout_frames = 15;
    c->min_speech_frames = 3;
}

void afo_vad_init(afo_vad* v, const afo_vad_config* c)
{   /* vad.rs:80-88 */
    v->cfg = *c;
    v->smoothed_energy = 0.0f;
    v->silence_frames = 0;
    v->speech_frames = 0;
    v->state = AFO_VAD_SILENCE;
}

void afo_vad_reset(afo_vad* v)
{   /* vad.rs:179-184 */
    v->smoothed_energy = 0.0f;
    v->silence_frames = 0;
    v->speech_frames = 0;
    v->state = AFO_VAD_SILENCE;
}

int afo_vad_step_energy(afo_vad* v, float energy)
{
    /* vad.rs:102-104: EMA, two f32 products and one f32 sum */
    float old_smoothed = v->smoothed_energy;
    volatile float a = v->cfg.smoothing_factor * energy;
    volatile float one_minus = 1.0f - v->cfg.smoothing_factor;
    volatile float b = one_minus * old_smoothed;
    v->smoothed_energy = a + b;

    /* vad.rs:108-112 */
    float detection = v->cfg.smoothing_factor > 0.0f ? v->smoothed_energy : energy;
    float dbfs = afo_energy_to_dbfs(detection);          /* vad.rs:115 */
    int is_speech = dbfs > v->cfg.threshold_db;          /* vad.rs:118 */

    switch (v->state) {                                  /* vad.rs:121-151 */
    case AFO_VAD_SILENCE:
        if (is_speech) {
            v->speech_frames = 1;
            v->silence_frames = 0;
            v->state = AFO_VAD_SPEECH;
        }
        break;
    case AFO_VAD_SPEECH:
        if (is_speech) {
            v->speech_frames += 1;
            v->silence_frames = 0;
        } else {
            v->silence_frames += 1;
            if (v->silence_frames >= v->cfg.silence_timeout_frames) {
                if (v->speech_frames >= v->cfg.min_speech_frames)
                    v->state = AFO_VAD_ENDING;
                else
                    v->state = AFO_VAD_SILENCE;
                v->speech_frames = 0;
            }
        }
        break;
    default: /* AFO_VAD_ENDING, vad.rs:146-150 */
        v->state = AFO_VAD_SILENCE;
        v->silence_frames = 0;
        break;
    }
    return v->state;
}

int afo_vad_detect(afo_vad* v, const float* frame, size_t n)
{   /* vad.rs:97-99 */
    return afo_vad_step_energy(v, afo_vad_energy(frame, n));
}

/* n_clips independent detectors (vad.rs:60-70: each owns its state), each fed its row of energies in order */
void afo_vad_scan_batch(const afo_vad_config* c, const float* energy, size_t n_clips, size_t n_frames,
                        unsigned char* states)
{
    for (size_t i = 0; i < n_clips; ++i) {
        afo_vad v;
        afo_vad_init(&v, c);
        for (size_t f = 0; f < n_frames; ++f)
            states[i * n_frames + f] = (unsigned char)afo_vad_step_energy(&v, energy[i * n_frames + f]);
    }
}

/* ---- ring buffer, capture.rs:84-161 ---------------------------------- */
int afo_ring_init(afo_ring* r, size_t capacity)
{
    r->buf = (float*)calloc(capacity ? capacity : 1, sizeof(float));
    r->capacity = capacity;
    r->write_pos = r->read_pos = 0;
    return r->buf ? 0 : -1;
}
void afo_ring_free(afo_ring* r) { free(r->buf); r->buf = NULL; }

size_t afo_ring_available(const afo_ring* r)
{   /* capture.rs:148-152 */
    if (r->capacity == 0) return 0;
    int64_t d = (int64_t)r->write_pos - (int64_t)r->read_pos + (int64_t)r->capacity;
    return (size_t)d % r->capacity;
}

size_t afo_ring_write(afo_ring* r, const float* data, size_t n)
{   /* capture.rs:103-122: free = capacity - used; writable = free - 1 */
    if (r->capacity == 0) return 0;
    size_t used = afo_ring_available(r);
    size_t avail = r->capacity > used ? r->capacity - used : 0;
    size_t room = avail > 0 ? avail - 1 : 0;
    size_t to_write = n < room ? n : room;
    for (size_t i = 0; i < to_write; ++i)
        r->buf[(r->write_pos + i) % r->capacity] = data[i];
    if (to_write > 0) r->write_pos = (r->write_pos + to_write) % r->capacity;
    return to_write;
}

size_t afo_ring_read(afo_ring* r, size_t size, float* out)
{   /* capture.rs:125-145 */
    size_t available = afo_ring_available(r);
    if (available == 0) return (size_t)-1;          /* None */
    size_t to_read = size < available ? size : available;
    for (size_t i = 0; i < to_read; ++i)
        out[i] = r->buf[(r->read_pos + i) % r->capacity];
    r->read_pos = (r->read_pos + to_read) % r->capacity;
    return to_read;
}

void afo_ring_clear(afo_ring* r)
{   /* capture.rs:155-160 */
    r->write_pos = r->read_pos = 0;
    memset(r->buf, 0, r->capacity * sizeof(float));
}

/* resampler.rs:132-147 */
void afo_chunker_process(size_t carry_in, size_t n_in, size_t chunk,
                         size_t* n_chunks, size_t* carry_out)
{
    if (chunk == 0) { *n_chunks = 0; *carry_out = 0; return; }
    size_t total = carry_in + n_in;
    *n_chunks = total / chunk;
    *carry_out = total % chunk;
}

/* ====================================================================== */
/* [B] build-defined ops, float64 (docs/OPS.md)                           */
/* ====================================================================== */

static uint32_t gcd_u32(uint32_t a, uint32_t b)
{
    while (b) { uint32_t t = a % b; a = b; b = t; }
    return a;
}

void afo_resample_ratio(uint32_t in_hz, uint32_t out_hz, uint32_t* L, uint32_t* M)
{
    uint32_t g = gcd_u32(in_hz, out_hz);
    if (g == 0) { *L = *M = 0; return; }
    *L = out_hz / g;
    *M = in_hz / g;
}

size_t afo_resample_out_len(size_t n_in, uint32_t L, uint32_t M)
{
    if (M == 0) return 0;
    unsigned __int128 p = (unsigned __int128)n_in * L;
    return (size_t)((p + M - 1) / M);
}

void afo_resample_design(uint32_t L, uint32_t M, uint32_t T, double* h)
{
    const size_t LT = (size_t)L * T;
    const double c = (double)(LT / 2);
    const double fcu = 0.5 / (double)(L > M ? L : M); /* cycles / upsampled sample */
    static const double a0 = 0.35875, a1 = 0.48829, a2 = 0.14128, a3 = 0.01168;
    for (size_t p = 0; p < LT; ++p) {
        double t = (double)p - c;
        double x = 2.0 * fcu * t;
        double s = (t == 0.0) ? 1.0 : sin(M_PI * x) / (M_PI * x);
        double ph = 2.0 * M_PI * (double)p / (double)LT;
        double w = a0 - a1 * cos(ph) + a2 * cos(2.0 * ph) - a3 * cos(3.0 * ph);
        h[p] = s * w;
    }
    for (uint32_t r = 0; r < L; ++r) {          /* unit DC gain per phase */
        double sum = 0.0;
        for (uint32_t k = 0; k < T; ++k) sum += h[r + (size_t)k * L];
        for (uint32_t k = 0; k < T; ++k) h[r + (size_t)k * L] /= sum;
    }
}

void afo_resample_f64(const float* x, size_t n, const double* h,
                      uint32_t L, uint32_t M, uint32_t T,
                      size_t m0, size_t n_out, double* y)
{
    for (size_t j = 0; j < n_out; ++j) {
        unsigned __int128 mm = (unsigned __int128)(m0 + j) * M;
        int64_t q = (int64_t)(mm / L);
        uint32_t r = (uint32_t)(mm % L);
        double acc = 0.0;
        for (uint32_t k = 0; k < T; ++k) {
            int64_t i = q - (int64_t)k + (int64_t)(T / 2);
            if (i < 0 || (uint64_t)i >= n) continue;
            acc += h[r + (size_t)k * L] * (double)x[i];
        }
        y[j] = acc;
    }
}

void afo_fir_f64(const float* x, size_t n, const float* h, size_t K, double* y)
{
    for (size_t i = 0; i < n; ++i) {
        double acc = 0.0;
        size_t kmax = K < i + 1 ? K : i + 1;
        for (size_t k = 0; k < kmax; ++k) acc += (double)h[k] * (double)x[i - k];
        y[i] = acc;
    }
}

size_t afo_stft_n_frames(size_t n, size_t n_fft, size_t hop)
{
    if (n < n_fft || hop == 0) return 0;
    return 1 + (n - n_fft) / hop;
}

void afo_hann_periodic(size_t N, double* w)
{
    for (size_t i = 0; i < N; ++i)
        w[i] = 0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)N);
}

/* in-place iterative radix-2 DIT, float64 */
static void fft_f64(double* re, double* im, size_t N)
{
    for (size_t i = 1, j = 0; i < N; ++i) {
        size_t bit = N >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (size_t len = 2; len <= N; len <<= 1) {
        size_t half = len >> 1;
        for (size_t k = 0; k < half; ++k) {
            double ang = -2.0 * M_PI * (double)k / (double)len;
            double wr = cos(ang), wi = sin(ang);
            for (size_t i = k; i < N; i += len) {
                size_t j = i + half;
                double tr = re[j] * wr - im[j] * wi;
                double ti = re[j] * wi + im[j] * wr;
                re[j] = re[i] - tr; im[j] = im[i] - ti;
                re[i] += tr;        im[i] += ti;
            }
        }
    }
}

void afo_stft_power_f64(const float* x, size_t n, size_t n_fft, size_t hop,
                        double* power)
{
    size_t nf = afo_stft_n_frames(n, n_fft, hop);
    size_t nb = n_fft / 2 + 1;
    double* w = (double*)malloc(n_fft * sizeof(double));
    double* re = (double*)malloc(n_fft * sizeof(double));
    double* im = (double*)malloc(n_fft * sizeof(double));
    afo_hann_periodic(n_fft, w);
    if ((n_fft & (n_fft - 1)) != 0) {
        /* a frame length that is not a power of two (400 = 25 ms at 16 kHz ...): the DFT sum itself, float64,
         * angles from a table indexed by k i mod n_fft (the same reduction as afo_dft_power_naive_f64) */
        double* c = (double*)malloc(n_fft * sizeof(double));
        double* s = (double*)malloc(n_fft * sizeof(double));
        for (size_t r = 0; r < n_fft; ++r) {
            double ang = -2.0 * M_PI * (double)r / (double)n_fft;
            c[r] = cos(ang); s[r] = sin(ang);
        }
        for (size_t f = 0; f < nf; ++f) {
            const float* xf = x + f * hop;
            for (size_t i = 0; i < n_fft; ++i) re[i] = w[i] * (double)xf[i];
            for (size_t k = 0; k < nb; ++k) {
                double sr = 0.0, si = 0.0;
                size_t ki = 0;
                for (size_t i = 0; i < n_fft; ++i) {
                    sr += re[i] * c[ki];
                    si += re[i] * s[ki];
                    ki += k; if (ki >= n_fft) ki -= n_fft;
                }
                power[f * nb + k] = sr * sr + si * si;
            }
        }
        free(c); free(s); free(w); free(re); free(im);
        return;
    }
    for (size_t f = 0; f < nf; ++f) {
        const float* xf = x + f * hop;
        for (size_t i = 0; i < n_fft; ++i) { re[i] = w[i] * (double)xf[i]; im[i] = 0.0; }
        fft_f64(re, im, n_fft);
        for (size_t k = 0; k < nb; ++k)
            power[f * nb + k] = re[k] * re[k] + im[k] * im[k];
    }
    free(w); free(re); free(im);
}

void afo_dft_power_naive_f64(const float* x, size_t n_fft, const double* w,
                             double* power)
{
    size_t nb = n_fft / 2 + 1;
    for (size_t k = 0; k < nb; ++k) {
        double sr = 0.0, si = 0.0;
        for (size_t i = 0; i < n_fft; ++i) {
            /* reduce k*i mod n_fft before scaling: keeps the angle exact */
            size_t ki = (k * i) % n_fft;
            double ang = -2.0 * M_PI * (double)ki / (double)n_fft;
            double v = w[i] * (double)x[i];
            sr += v * cos(ang);
            si += v * sin(ang);
        }
        power[k] = sr * sr + si * si;
    }
}

void afo_level_meter_f64(const float* x, size_t n, double* rms, double* peak, double* dbfs)
{
    double s = 0.0, m = 0.0;
    for (size_t i = 0; i < n; ++i) {
        double v = (double)x[i];
        s += v * v;
        if (fabs(v) > m) m = fabs(v);
    }
    *rms = n ? sqrt(s / (double)n) : 0.0;
    *peak = m;
    *dbfs = *rms > 0.0 ? 20.0 * log10(*rms) : -INFINITY;
}

double afo_hz_to_mel_htk(double hz) { return 2595.0 * log10(1.0 + hz / 700.0); }
double afo_mel_to_hz_htk(double mel) { return 700.0 * (pow(10.0, mel / 2595.0) - 1.0); }

/* docs/OPS.md "Mel filterbank": the Slaney scale is linear (200/3 Hz per mel) below 1 kHz and
 * logarithmic (27 mels per factor 6.4) above. */
static double hz_to_mel_any(double hz, int slaney)
{
    if (!slaney) return afo_hz_to_mel_htk(hz);
    const double f_sp = 200.0 / 3.0, brk = 1000.0, logstep = log(6.4) / 27.0;
    return hz >= brk ? brk / f_sp + log(hz / brk) / logstep : hz / f_sp;
}
static double mel_to_hz_any(double mel, int slaney)
{
    if (!slaney) return afo_mel_to_hz_htk(mel);
    const double f_sp = 200.0 / 3.0, brk = 1000.0, logstep = log(6.4) / 27.0;
    return mel >= brk / f_sp ? brk * exp(logstep * (mel - brk / f_sp)) : f_sp * mel;
}

void afo_mel_weights_ex(double sample_rate, size_t n_fft, size_t n_mels, double f_min,
                        double f_max, int scale, int norm, double* W)
{
    size_t nb = n_fft / 2 + 1;
    double* edges = (double*)malloc((n_mels + 2) * sizeof(double));
    double m_lo = hz_to_mel_any(f_min, scale), m_hi = hz_to_mel_any(f_max, scale);
    for (size_t i = 0; i < n_mels + 2; ++i)
        edges[i] = mel_to_hz_any(m_lo + (m_hi - m_lo) * (double)i / (double)(n_mels + 1), scale);
    for (size_t m = 0; m < n_mels; ++m) {
        double lo = edges[m], ce = edges[m + 1], hi = edges[m + 2];
        for (size_t k = 0; k < nb; ++k) {
            double f = (double)k * sample_rate / (double)n_fft;
            double up = (f - lo) / (ce - lo);
            double dn = (hi - f) / (hi - ce);
            double v = up < dn ? up : dn;
            if (norm) v *= 2.0 / (hi - lo);
            W[m * nb + k] = v > 0.0 ? v : 0.0;
        }
    }
    free(edges);
}

void afo_mel_weights(double sample_rate, size_t n_fft, size_t n_mels,
                     double f_min, double f_max, double* W)
{
    afo_mel_weights_ex(sample_rate, n_fft, n_mels, f_min, f_max, 0, 0, W);
}

void afo_mel_apply_f64(const double* power, size_t n_frames, size_t n_bins,
                       const double* W, size_t n_mels, double* mel)
{
    for (size_t f = 0; f < n_frames; ++f)
        for (size_t m = 0; m < n_mels; ++m) {
            double acc = 0.0;
            for (size_t k = 0; k < n_bins; ++k)
                acc += W[m * n_bins + k] * power[f * n_bins + k];
            mel[f * n_mels + m] = acc;
        }
}

size_t afo_features_f64(const float* pcm, size_t n_frames_pcm, unsigned channels,
                        double sample_rate, size_t n_fft, size_t hop,
                        size_t n_mels, double f_min, double f_max, double* mel)
{
    size_t nb = n_fft / 2 + 1;
    float* mono = (float*)malloc((n_frames_pcm ? n_frames_pcm : 1) * sizeof(float));
    size_t n = afo_to_mono(pcm, n_frames_pcm * channels, channels, mono);
    size_t nf = afo_stft_n_frames(n, n_fft, hop);
    if (nf) {
        double* power = (double*)malloc(nf * nb * sizeof(double));
        double* W = (double*)malloc(n_mels * nb * sizeof(double));
        afo_stft_power_f64(mono, n, n_fft, hop, power);
        afo_mel_weights(sample_rate, n_fft, n_mels, f_min, f_max, W);
        afo_mel_apply_f64(power, nf, nb, W, n_mels, mel);
        free(power); free(W);
    }
    free(mono);
    return nf;
}

/* [B, parity unpinned] Cubic interpolation mode (docs/OPS.md): the configuration of resampler.rs:43-49 (rubato
 * FastFixedIn, PolynomialDegree::Cubic) as a closed formula -- the cubic through (-1, x[i-1]), (0, x[i]), (1, x[i+1]),
 * (2, x[i+2]) at i + t = m * in_hz / out_hz, zero extension, n_out outputs.  float64 throughout. */
void afo_resample_cubic(const float* x, size_t n, unsigned in_hz, unsigned out_hz, size_t n_out, double* y)
{
    for (size_t m = 0; m < n_out; ++m) {
        const double pos = (double)m * ((double)in_hz / (double)out_hz);
        const long long i = (long long)pos;
        const double t = pos - (double)i;
        double v[4];
        for (int k = 0; k < 4; ++k) {
            const long long j = i - 1 + k;
            v[k] = (j >= 0 && (size_t)j < n) ? (double)x[j] : 0.0;
        }
        const double a1 = -v[0] / 3.0 - v[1] / 2.0 + v[2] - v[3] / 6.0;
        const double a2 = (v[0] + v[2]) / 2.0 - v[1];
        const double a3 = (v[1] - v[2]) / 2.0 + (v[3] - v[0]) / 6.0;
        y[m] = v[1] + t * (a1 + t * (a2 + t * a3));
    }
}
