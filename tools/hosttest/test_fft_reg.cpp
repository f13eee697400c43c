// Host-side check of the register butterflies (ss_fft_reg.h is __host__ __device__): every radix against a naive f64 DFT.
// build + run (no GPU needed): hipcc -O2 -std=c++17 -I mfcc-rust_amd/csrc tools/hosttest/test_fft_reg.cpp -o /tmp/test_fft_reg && /tmp/test_fft_reg
#include "ss_fft_reg.h"

#include <cmath>
#include <cstdio>
#include <random>

template <int R>
static double check()
{
    std::mt19937 g(R);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    double worst = 0;
    for (int rep = 0; rep < 50; ++rep) {
        float2 v[R];
        double xr[R], xi[R];
        for (int i = 0; i < R; ++i) {
            v[i] = make_float2(u(g), u(g));
            xr[i] = v[i].x;
            xi[i] = v[i].y;
        }
        ss::fft_reg<R>(v);
        for (int k = 0; k < R; ++k) {
            double sr = 0, si = 0;
            for (int n = 0; n < R; ++n) {
                const double a = -2.0 * M_PI * n * k / R;
                sr += xr[n] * std::cos(a) - xi[n] * std::sin(a);
                si += xr[n] * std::sin(a) + xi[n] * std::cos(a);
            }
            worst = std::fmax(worst, std::fmax(std::fabs(sr - v[k].x), std::fabs(si - v[k].y)));
        }
    }
    std::printf("radix %2d: max abs err %.3g\n", R, worst);
    return worst;
}

int main()
{
    double w = 0;
    w = std::fmax(w, check<2>());
    w = std::fmax(w, check<4>());
    w = std::fmax(w, check<8>());
    w = std::fmax(w, check<16>());
    w = std::fmax(w, check<32>());
    return w < 2e-5 ? 0 : 1;
}
