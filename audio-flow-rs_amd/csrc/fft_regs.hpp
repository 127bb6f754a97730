// fft_regs.hpp -- small power-of-two complex FFTs on register-resident arrays, written so that
// every twiddle butterfly is FMA-shaped (Linzer-Feig: the cosine, or the sine where it is the
// larger of the two, is factored out of W*B and applied by the output FMAs):
//     general twiddle   6 instructions per radix-2 butterfly (2 FMA for u = W*B / scale, 4 FMA for A +- scale*u)
//     W = 1, W = -i     4 instructions
// A 16-point transform costs 148, a 32-point one 388 VALU instructions (decimation in frequency
// with separate multiplies and adds: 188 / 456).
//
// Decimation in time: element e of the array must hold input brev(e) on entry; on return
// element k holds X[k] = sum_n x[n] e^{-2 pi i n k / N}.  Everything is indexed at compile time
// after unrolling.  The header is plain C++ (tests/cpp/fft_regs_test.cpp compiles it on the host).
#pragma once

#if defined(__HIPCC__)
#define AF_FFT_HD __device__ __forceinline__
#define AF_FFT_CONST __device__ static constexpr
#else
#define AF_FFT_HD inline
#define AF_FFT_CONST static constexpr
#endif

namespace af {
namespace fftr {

// e^{-2 pi i j / 32} = kC[j] - i kS[j], j = 0..15;  kR = tan where |cos| >= |sin| (kForm 2),
// cot where |sin| > |cos| (kForm 3);  kForm 0: W = 1, kForm 1: W = -i
AF_FFT_CONST float kC[16] = {1.0f, 0.980785251f, 0.923879504f, 0.831469595f, 0.707106769f, 0.555570245f,
                             0.382683426f, 0.195090324f, 0.0f, -0.195090324f, -0.382683426f,
                             -0.555570245f, -0.707106769f, -0.831469595f, -0.923879504f, -0.980785251f};
AF_FFT_CONST float kS[16] = {0.0f, 0.195090324f, 0.382683426f, 0.555570245f, 0.707106769f, 0.831469595f,
                             0.923879504f, 0.980785251f, 1.0f, 0.980785251f, 0.923879504f, 0.831469595f,
                             0.707106769f, 0.555570245f, 0.382683426f, 0.195090324f};
AF_FFT_CONST float kR[16] = {0.0f, 0.198912367f, 0.414213568f, 0.668178618f, 1.0f, 0.668178618f,
                             0.414213568f, 0.198912367f, 0.0f, -0.198912367f, -0.414213568f,
                             -0.668178618f, -1.0f, -0.668178618f, -0.414213568f, -0.198912367f};
AF_FFT_CONST int kForm[16] = {0, 2, 2, 2, 2, 3, 3, 3, 1, 3, 3, 3, 2, 2, 2, 2};

AF_FFT_HD constexpr int brev(int x, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

AF_FFT_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// one radix-2 DIT butterfly, twiddle e^{-2 pi i T / 32}:  (A, B) <- (A + W B, A - W B).
// T is a compile-time constant after unrolling; the untaken branches fold away.
AF_FFT_HD void butterfly(const int T, float& ar, float& ai, float& br, float& bi)
{
    if (kForm[T] == 0) {
        const float xr = ar + br, xi = ai + bi;
        br = ar - br; bi = ai - bi;
        ar = xr; ai = xi;
    } else if (kForm[T] == 1) {                    // W B = bi - i br
        const float xr = ar + bi, xi = ai - br;
        const float yr = ar - bi, yi = ai + br;
        ar = xr; ai = xi; br = yr; bi = yi;
    } else if (kForm[T] == 2) {                    // W B = c [(br + bi t) + i (bi - br t)]
        const float t = kR[T], c = kC[T];
        const float ur = fma_(bi, t, br), ui = fma_(br, -t, bi);
        br = fma_(ur, -c, ar); bi = fma_(ui, -c, ai);
        ar = fma_(ur, c, ar); ai = fma_(ui, c, ai);
    } else {                                       // W B = s [(br q + bi) + i (bi q - br)]
        const float q = kR[T], s = kS[T];
        const float ur = fma_(br, q, bi), ui = fma_(bi, q, -br);
        br = fma_(ur, -s, ar); bi = fma_(ui, -s, ai);
        ar = fma_(ur, s, ar); ai = fma_(ui, s, ai);
    }
}

// stages SPAN0, 2 SPAN0, ... N/2 of an N-point DIT transform (SPAN0 = 1: the whole transform;
// SPAN0 = 2: the caller has done the first, twiddle-free stage itself, e.g. fused with a window)
template <int N, int SPAN>
AF_FFT_HD void dit_stage(float (&re)[N], float (&im)[N])
{
#pragma unroll
    for (int g = 0; g < N; g += 2 * SPAN) {
#pragma unroll
        for (int j = 0; j < SPAN; ++j)
            butterfly(j * (16 / SPAN), re[g + j], im[g + j], re[g + j + SPAN], im[g + j + SPAN]);
    }
}

template <int N, int SPAN0 = 1>
AF_FFT_HD void dit_fft(float (&re)[N], float (&im)[N])
{
    static_assert(N == 4 || N == 8 || N == 16 || N == 32, "register FFT sizes");
    if constexpr (SPAN0 <= 1) dit_stage<N, 1>(re, im);
    if constexpr (SPAN0 <= 2 && N >= 4) dit_stage<N, 2>(re, im);
    if constexpr (SPAN0 <= 4 && N >= 8) dit_stage<N, 4>(re, im);
    if constexpr (SPAN0 <= 8 && N >= 16) dit_stage<N, 8>(re, im);
    if constexpr (N >= 32) dit_stage<N, 16>(re, im);
}

}  // namespace fftr
}  // namespace af
