/*
 * af_oracle_fast.c -- float32 CPU port of the feature path, used ONLY as the
 * timed `cpu_baseline` leg of bench.py (kind "port") and cross-checked in
 * tests against the float64 oracle.  TEST INFRASTRUCTURE ONLY (af_oracle.h).
 *
 * The reference's audio path is scalar, single-threaded Rust with no SIMD
 * and no threads (SURVEY.md section 3 "hot loops"); threads==1 keeps that
 * character, threads>1 spreads whole clips over host cores with OpenMP.
 *
 * Compile with: gcc -O3 -march=x86-64-v3 -fopenmp
 */
#include "af_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

int afo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void afo_resample_f32(const float* x, size_t n, const float* h,
                      uint32_t L, uint32_t M, uint32_t T,
                      size_t m0, size_t n_out, float* y)
{
    for (size_t j = 0; j < n_out; ++j) {
        unsigned __int128 mm = (unsigned __int128)(m0 + j) * M;
        int64_t q = (int64_t)(mm / L);
        uint32_t r = (uint32_t)(mm % L);
        float acc = 0.0f;
        for (uint32_t k = 0; k < T; ++k) {
            int64_t i = q - (int64_t)k + (int64_t)(T / 2);
            if (i < 0 || (uint64_t)i >= n) continue;
            acc += h[r + (size_t)k * L] * x[i];
        }
        y[j] = acc;
    }
}

void afo_fir_f32(const float* x, size_t n, const float* h, size_t K, float* y)
{
    for (size_t i = 0; i < n; ++i) {
        float acc = 0.0f;
        size_t kmax = K < i + 1 ? K : i + 1;
        for (size_t k = 0; k < kmax; ++k) acc += h[k] * x[i - k];
        y[i] = acc;
    }
}

/* ---- real FFT of length N via complex FFT of length N/2, float32 ------- */
typedef struct {
    size_t N, H;           /* N real points, H = N/2 complex points */
    unsigned* rev;         /* bit reversal for H */
    float *twr, *twi;      /* e^{-2 pi i k / H}, k < H/2 */
    float *pr, *pi;        /* e^{-2 pi i k / N}, k <= H/2 (post-processing) */
    float* win;            /* periodic Hann */
} rfft_plan;

static void plan_init(rfft_plan* p, size_t N)
{
    p->N = N; p->H = N / 2;
    size_t H = p->H;
    p->rev = (unsigned*)malloc(H * sizeof(unsigned));
    unsigned bits = 0; while (((size_t)1 << bits) < H) ++bits;
    for (size_t i = 0; i < H; ++i) {
        unsigned r = 0;
        for (unsigned b = 0; b < bits; ++b) if (i & ((size_t)1 << b)) r |= 1u << (bits - 1 - b);
        p->rev[i] = r;
    }
    p->twr = (float*)malloc(H / 2 * sizeof(float));
    p->twi = (float*)malloc(H / 2 * sizeof(float));
    for (size_t k = 0; k < H / 2; ++k) {
        double a = -2.0 * M_PI * (double)k / (double)H;
        p->twr[k] = (float)cos(a); p->twi[k] = (float)sin(a);
    }
    p->pr = (float*)malloc((H / 2 + 1) * sizeof(float));
    p->pi = (float*)malloc((H / 2 + 1) * sizeof(float));
    for (size_t k = 0; k <= H / 2; ++k) {
        double a = -2.0 * M_PI * (double)k / (double)N;
        p->pr[k] = (float)cos(a); p->pi[k] = (float)sin(a);
    }
    p->win = (float*)malloc(N * sizeof(float));
    for (size_t i = 0; i < N; ++i)
        p->win[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)N));
}

static void plan_free(rfft_plan* p)
{
    free(p->rev); free(p->twr); free(p->twi); free(p->pr); free(p->pi); free(p->win);
}

/* power[k], k=0..N/2 of the Hann-windowed frame x[0..N) */
static void frame_power_f32(const rfft_plan* p, const float* x, float* zr, float* zi,
                            float* power)
{
    const size_t H = p->H;
    for (size_t i = 0; i < H; ++i) {
        unsigned r = p->rev[i];
        zr[r] = x[2 * i] * p->win[2 * i];
        zi[r] = x[2 * i + 1] * p->win[2 * i + 1];
    }
    for (size_t len = 2; len <= H; len <<= 1) {
        size_t half = len >> 1, step = H / len;
        for (size_t i = 0; i < H; i += len)
            for (size_t k = 0; k < half; ++k) {
                float wr = p->twr[k * step], wi = p->twi[k * step];
                size_t a = i + k, b = a + half;
                float tr = zr[b] * wr - zi[b] * wi;
                float ti = zr[b] * wi + zi[b] * wr;
                zr[b] = zr[a] - tr; zi[b] = zi[a] - ti;
                zr[a] += tr;        zi[a] += ti;
            }
    }
    /* X[k] = (Z[k]+conj(Z[H-k]))/2 + e^{-2 pi i k/N} (Z[k]-conj(Z[H-k]))/(2i) */
    power[0] = (zr[0] + zi[0]) * (zr[0] + zi[0]);
    power[H] = (zr[0] - zi[0]) * (zr[0] - zi[0]);
    for (size_t k = 1; k <= H / 2; ++k) {
        size_t j = H - k;
        float ar = 0.5f * (zr[k] + zr[j]), ai = 0.5f * (zi[k] - zi[j]);
        float br = 0.5f * (zi[k] + zi[j]), bi = -0.5f * (zr[k] - zr[j]);
        float wr = p->pr[k], wi = p->pi[k];
        float tr = br * wr - bi * wi, ti = br * wi + bi * wr;
        float xr = ar + tr, xi = ai + ti;      /* X[k]   */
        float yr = ar - tr, yi = -(ai - ti);   /* X[H-k] */
        power[k] = xr * xr + xi * xi;
        power[j] = yr * yr + yi * yi;
    }
}

size_t afo_features_f32_batch(const float* pcm, size_t n_clips,
                              size_t n_frames_pcm, unsigned channels,
                              double sample_rate, size_t n_fft, size_t hop,
                              size_t n_mels, double f_min, double f_max,
                              float* mel, int threads)
{
    const size_t nb = n_fft / 2 + 1;
    const size_t nf = afo_stft_n_frames(n_frames_pcm, n_fft, hop);
    if (nf == 0) return 0;

    rfft_plan plan; plan_init(&plan, n_fft);
    double* Wd = (double*)malloc(n_mels * nb * sizeof(double));
    afo_mel_weights(sample_rate, n_fft, n_mels, f_min, f_max, Wd);
    float* W = (float*)malloc(n_mels * nb * sizeof(float));
    size_t* lo = (size_t*)malloc(n_mels * sizeof(size_t));
    size_t* hi = (size_t*)malloc(n_mels * sizeof(size_t));
    for (size_t m = 0; m < n_mels; ++m) {
        lo[m] = nb; hi[m] = 0;
        for (size_t k = 0; k < nb; ++k) {
            W[m * nb + k] = (float)Wd[m * nb + k];
            if (Wd[m * nb + k] != 0.0) { if (k < lo[m]) lo[m] = k; hi[m] = k + 1; }
        }
        if (lo[m] > hi[m]) lo[m] = hi[m] = 0;
    }
    free(Wd);

    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
        float* mono = (float*)malloc(n_frames_pcm * sizeof(float));
        float* zr = (float*)malloc(n_fft / 2 * sizeof(float));
        float* zi = (float*)malloc(n_fft / 2 * sizeof(float));
        float* power = (float*)malloc(nb * sizeof(float));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (long c = 0; c < (long)n_clips; ++c) {
            const float* src = pcm + (size_t)c * n_frames_pcm * channels;
            if (channels == 1) {
                memcpy(mono, src, n_frames_pcm * sizeof(float));
            } else {
                for (size_t i = 0; i < n_frames_pcm; ++i) {
                    float s = 0.0f;
                    for (unsigned ch = 0; ch < channels; ++ch) s += src[i * channels + ch];
                    mono[i] = s / (float)channels;
                }
            }
            float* out = mel + (size_t)c * nf * n_mels;
            for (size_t f = 0; f < nf; ++f) {
                frame_power_f32(&plan, mono + f * hop, zr, zi, power);
                for (size_t m = 0; m < n_mels; ++m) {
                    float acc = 0.0f;
                    for (size_t k = lo[m]; k < hi[m]; ++k) acc += W[m * nb + k] * power[k];
                    out[f * n_mels + m] = acc;
                }
            }
        }
        free(mono); free(zr); free(zi); free(power);
    }
    free(W); free(lo); free(hi);
    plan_free(&plan);
    return nf;
}
