// Single-utterance latency of the C ABI itself (what a Rust / C++ caller of ss_mfcc pays), without the Python front:
//   g++ -O2 -std=c++17 -Iinclude tools/abi_latency.cpp -Lmfcc-rust_amd/lib -lspeechsauce_amd -Wl,-rpath,$PWD/mfcc-rust_amd/lib -o ab/abi_latency && ab/abi_latency
#include "speechsauce_amd.h"

#include <chrono>
#include <cstdio>
#include <vector>

int main()
{
    ss_params p;
    ss_params_default(&p, 16000);
    ss_config *cfg = nullptr;
    if (ss_config_create(&p, &cfg) != 0) {
        std::printf("ss_config_create: %s\n", ss_last_error_string());
        return 1;
    }
    for (size_t secs : {1, 3, 10}) {
        const size_t n = 16000 * secs;
        std::vector<float> x(n);
        for (size_t i = 0; i < n; ++i) x[i] = 0.1f * static_cast<float>((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.05f;
        size_t T = 0;
        ss_num_frames(&p, n, &T);
        std::vector<float> out(T * p.num_cepstral);
        for (int i = 0; i < 100; ++i) ss_mfcc(cfg, x.data(), n, out.data());
        const auto t0 = std::chrono::steady_clock::now();
        const int reps = 2000;
        for (int i = 0; i < reps; ++i) ss_mfcc(cfg, x.data(), n, out.data());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        std::printf("ss_mfcc, one %zu s clip (%zu frames): %.1f us per call, out[0] = %g\n", secs, T, us, out[0]);
    }
    ss_config_destroy(cfg);
    return 0;
}
