// Host check of audio-flow-rs_amd/csrc/fft_regs.hpp: the register FFTs against a float64 DFT.
// Prints the worst relative error per size; exit code 1 above 2e-6.
#include <cmath>
#include <complex>
#include <cstdio>
#include <random>

#include "../../audio-flow-rs_amd/csrc/fft_regs.hpp"

template <int N, int BITS>
static double run()
{
    std::mt19937 rng(7 + N);
    std::uniform_real_distribution<float> d(-1.0f, 1.0f);
    double worst = 0.0;
    for (int rep = 0; rep < 50; ++rep) {
        float xr[N], xi[N], re[N], im[N];
        for (int n = 0; n < N; ++n) { xr[n] = d(rng); xi[n] = d(rng); }
        for (int e = 0; e < N; ++e) { re[e] = xr[af::fftr::brev(e, BITS)]; im[e] = xi[af::fftr::brev(e, BITS)]; }
        af::fftr::dit_fft<N>(re, im);
        double num = 0.0, den = 0.0;
        for (int k = 0; k < N; ++k) {
            std::complex<double> s = 0.0;
            for (int n = 0; n < N; ++n)
                s += std::complex<double>(xr[n], xi[n]) * std::polar(1.0, -2.0 * M_PI * n * k / N);
            num += std::norm(s - std::complex<double>(re[k], im[k]));
            den += std::norm(s);
        }
        worst = std::fmax(worst, std::sqrt(num / den));
    }
    std::printf("N=%d rel err %.3e\n", N, worst);
    return worst;
}

int main()
{
    double w = 0.0;
    w = std::fmax(w, run<4, 2>());
    w = std::fmax(w, run<8, 3>());
    w = std::fmax(w, run<16, 4>());
    w = std::fmax(w, run<32, 5>());
    return w < 2e-6 ? 0 : 1;
}
