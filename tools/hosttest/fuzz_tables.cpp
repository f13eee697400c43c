// Host-side property test of the kernels' table builders (no GPU), meant to run under AddressSanitizer + UBSan:
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all -Iinclude -Imfcc-rust_amd/csrc \
//       tools/hosttest/fuzz_tables.cpp mfcc-rust_amd/csrc/ss_host.cpp -o /tmp/fuzz_tables && /tmp/fuzz_tables [cases] [seed]
// For pseudo-random valid configurations (sample rate, fft_points, frame length, filter / coefficient counts, band edges, mel
// scale / norm, window) it builds the host tables and every kernel's LDS block and checks what the kernels rely on:
//   * every filter of the bank sits on exactly one (slot, lane) and the weights the lane will multiply -- its row of the mel
//     block, starting at the P bin in the start table -- are, bit for bit, the dense bank's row (zero outside);
//   * no (slot, lane) reads past the P row the kernel keeps;
//   * the cosine rows equal the host DCT table in the layout the kernel indexes (twice-folded per-lane rows of the 4096 kernel).
#include "ss_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static unsigned long long g_state = 1;
static unsigned rnd()
{
    g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
    return static_cast<unsigned>(g_state >> 33);
}
static double urand() { return rnd() / 2147483648.0; }

static int g_fail = 0;
#define CHECK(cond, ...)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            if (g_fail < 20) {                            \
                std::printf("FAIL %s: ", what.c_str());   \
                std::printf(__VA_ARGS__);                 \
                std::printf("\n");                        \
            }                                             \
            ++g_fail;                                     \
        }                                                 \
    } while (0)

// mel block check shared by all kernels: start / filt tables [S][LANES], weight rows [LANES][wpitch] at melw0
static void check_mel_block(const std::string &what, const ss::HostTables &t, const std::vector<float> &tab, int k_start, int k_filt, size_t melw0, int S,
                            int LANES, const int32_t *q4, int wpitch, int prow_bins)
{
    const size_t M = t.params.num_filters, F = t.d.n_bins;
    // k_filt < 0: one packed word per (slot, lane) at k_start -- first bin | filter index << 16 (the 4096-point layout)
    std::vector<int32_t> start_v(static_cast<size_t>(S) * LANES), filt_v(static_cast<size_t>(S) * LANES);
    for (size_t q = 0; q < start_v.size(); ++q) {
        const int32_t w = reinterpret_cast<const int32_t *>(tab.data() + k_start)[q];
        start_v[q] = k_filt < 0 ? (w & 0xffff) : w;
        filt_v[q] = k_filt < 0 ? (w >> 16) : reinterpret_cast<const int32_t *>(tab.data() + k_filt)[q];
    }
    const int32_t *start = start_v.data(), *filt = filt_v.data();
    CHECK(melw0 + static_cast<size_t>(LANES) * wpitch <= tab.size(), "mel block past the table (%zu + %d x %d > %zu)", melw0, LANES, wpitch, tab.size());
    if (melw0 + static_cast<size_t>(LANES) * wpitch > tab.size()) return;
    std::vector<int> seen(M, 0);
    int off = 0;
    for (int s = 0; s < S; ++s) {
        const int span = 4 * q4[s];
        CHECK(off + span <= wpitch, "slot %d: taps %d..%d past the row pitch %d", s, off, off + span, wpitch);
        for (int j = 0; j < LANES; ++j) {
            const int st = start[s * LANES + j], m = filt[s * LANES + j];
            CHECK(st >= 0 && st + span <= prow_bins, "slot %d lane %d reads P bins %d..%d of %d", s, j, st, st + span, prow_bins);
            CHECK(m >= -1 && m < static_cast<int>(M), "slot %d lane %d: filter index %d", s, j, m);
            if (m < 0 || m >= static_cast<int>(M) || st < 0) continue;
            ++seen[m];
            const float *row = tab.data() + melw0 + static_cast<size_t>(j) * wpitch + off;
            for (size_t b = 0; b < F; ++b) {
                const float want = t.fb_dense[m * F + b];
                const long i = static_cast<long>(b) - st;
                const float got = i >= 0 && i < span ? row[i] : 0.f;
                if (std::memcmp(&want, &got, 4) != 0 && !(want == 0.f && got == 0.f)) {
                    CHECK(false, "filter %d bin %zu: table %.9g, bank %.9g (slot %d lane %d start %d)", m, b, got, want, s, j, st);
                    break;
                }
            }
        }
        off += span;
    }
    for (size_t m = 0; m < M; ++m) CHECK(seen[m] == 1, "filter %zu placed %d times", m, seen[m]);
}

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? std::atoi(argv[1]) : 600;
    g_state = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 12345;
    static const unsigned rates[] = {8000, 11025, 16000, 22050, 32000, 44100, 48000};
    static const unsigned ffts[] = {256, 512, 512, 1024, 2048, 2048, 4096, 4096, 400, 128, 8192};
    int built[10] = {0};
    int valid = 0;
    for (int it = 0; it < cases; ++it) {
        ss_params p;
        const unsigned sr = rates[rnd() % 7];
        ss_params_default(&p, sr);
        p.fft_points = ffts[rnd() % 11];
        const double fl = (0.25 + 0.75 * urand()) * p.fft_points / sr;
        p.frame_length = static_cast<float>(rnd() % 4 == 0 ? static_cast<double>(p.fft_points) / sr : fl);
        p.frame_stride = static_cast<float>(p.frame_length * (rnd() % 3 == 0 ? 1.0 : 0.1 + 0.6 * urand()));
        static const unsigned fcounts[] = {1, 2, 5, 13, 20, 23, 26, 32, 40, 40, 48, 64, 80, 100, 101, 128, 128, 130, 200, 252, 255, 256, 256, 300};
        p.num_filters = fcounts[rnd() % 24];
        p.num_cepstral = 1 + rnd() % (p.num_filters < 64 ? p.num_filters : 64);
        if (rnd() % 3 == 0) p.low_frequency = static_cast<float>(urand() * sr / 8);
        if (rnd() % 2 == 0) p.high_frequency = static_cast<float>(sr / 2.0 * (0.3 + 0.7 * urand()));
        p.mel_scale = rnd() % 4 == 0 ? 1 + rnd() % 2 : 0;
        p.mel_norm = p.mel_scale && rnd() % 2 ? 1 : 0;
        p.mfcc_window = rnd() % 3;
        p.spectrum_exponent = 1 + rnd() % 2;
        p.dc_elimination = rnd() % 2;
        p.framing = rnd() % 8 == 0 ? SS_FRAMING_CENTER : SS_FRAMING_CONTRACT;
        ss::HostTables t;
        if (ss::build_tables(p, t) != 0) continue;  // rejected configurations (band edges, frame longer than fft_points ...) are fine
        ++valid;
        char name[256];
        std::snprintf(name, sizeof name, "case %d (sr %u fft %u flen %u step %u M %u C %u lo %.1f hi %.1f scale %d norm %d win %d)", it, sr, p.fft_points, t.d.flen,
                      t.d.step, p.num_filters, p.num_cepstral, p.low_frequency, p.high_frequency, p.mel_scale, p.mel_norm, p.mfcc_window);
        const size_t M = p.num_filters, Cc = p.num_cepstral;
        {
            ss::Fast512Tables f;
            ss::build_fast512(t, f);
            if (f.ok) {
                ++built[0];
                namespace L = ss::fast512_layout;
                check_mel_block(std::string(name) + " fast512", t, f.tab, L::kStart, L::kFilt, L::kMelW, 3, 16, f.q4, f.wpitch, f.fullp ? 260 : 132);
            }
        }
        {
            ss::Mfcc512wTables f;
            ss::build_mfcc512w(t, f);
            if (f.ok) {
                ++built[1];
                namespace L = ss::mfcc512w_layout;
                check_mel_block(std::string(name) + " mfcc512w", t, f.tab, L::kStart, L::kFilt, L::kMelW, 5, 16, f.q4, f.wpitch, 260);
            }
        }
        {
            ss::Mel512Tables f;
            ss::build_mel512(t, f);
            if (f.ok) {
                ++built[2];
                namespace L = ss::mel512_layout;
                check_mel_block(std::string(name) + " mel512", t, f.tab, L::kStart, L::kFilt, L::kMelW, 5, 16, f.q4, f.wpitch, f.fullp ? 260 : 132);
            }
        }
        {
            ss::Mfcc256Tables f;
            ss::build_mfcc256(t, f);
            if (f.ok) {
                ++built[3];
                namespace L = ss::mfcc256_layout;
                check_mel_block(std::string(name) + " mfcc256", t, f.tab, L::kStart, L::kFilt, L::kMelW, 3, 16, f.q4, f.wpitch, 132);
            }
        }
        for (int mel = 0; mel < 2; ++mel) {
            ss::Mfcc1024Tables f;
            if (mel) ss::build_mel1024(t, f);
            else ss::build_mfcc1024(t, f);
            if (f.ok) {
                ++built[4 + mel];
                namespace L = ss::mfcc1024_layout;
                check_mel_block(std::string(name) + (mel ? " mel1024" : " mfcc1024"), t, f.tab, L::kStart, L::kFilt, L::kMelW, 4, 32, f.q4, f.wpitch, f.fullp ? 516 : 260);
            }
        }
        {
            ss::Mfcc2048Tables f;
            ss::build_mfcc2048(t, f);
            if (f.ok) {
                ++built[6];
                namespace L = ss::mfcc2048_layout;
                check_mel_block(std::string(name) + " mfcc2048", t, f.tab, L::kStart, L::kFilt, L::kMelW, 4, 32, f.q4, f.wpitch, f.fullp ? 1028 : 516);
            }
        }
        {
            ss::Mel2048Tables f;
            ss::build_mel2048(t, f);
            if (f.ok) {
                ++built[7];
                namespace L = ss::mel2048_layout;
                check_mel_block(std::string(name) + " mel2048", t, f.tab, L::kStart, L::kFilt, L::kMelW, 4, 32, f.q4, f.wpitch, f.fullp ? 1028 : 516);
            }
        }
        for (int mel = 0; mel < 2; ++mel) {
            ss::Mfcc4096Tables f;
            if (mel) ss::build_mel4096(t, f);
            else ss::build_mfcc4096(t, f);
            if (!f.ok) continue;
            ++built[8 + mel];
            namespace L = ss::mfcc4096_layout;
            const std::string what = std::string(name) + (mel ? " mel4096" : " mfcc4096");
            check_mel_block(what, t, f.tab, L::kStart, -1, static_cast<size_t>(L::kCos) + f.cos_floats, 4, 64, f.q4, f.wpitch, 1028);
            if (mel) continue;
            // cosine block: what the kernel's DCT stage multiplies must be the host DCT table
            if (f.dct_fold2) {
                CHECK(M % 4 == 0 && Cc <= 43, "fold2 for M %zu C %zu", M, Cc);
                const size_t ne = (Cc + 1) / 2, no = Cc / 2, nep = (ne + 1) & ~size_t(1);
                CHECK(nep + 2 * no <= 64, "lanes: %zu + 2 x %zu", nep, no);
                std::vector<int> cover(Cc * (M / 2), 0);
                for (size_t lane = 0; lane < 64; ++lane) {
                    const float *row = f.tab.data() + L::kCos + lane * L::kCosLanePitch;
                    size_t c = 0, m0 = 0, n = 0;
                    if (lane < ne) {
                        c = 2 * lane, m0 = 0, n = M / 4;
                    } else if (lane >= nep && (lane - nep) / 2 < no) {
                        c = 2 * ((lane - nep) / 2) + 1, m0 = 64 * ((lane - nep) & 1);
                        n = M / 2 > m0 ? (M / 2 - m0 < 64 ? M / 2 - m0 : 64) : 0;
                    }
                    for (size_t i = 0; i < 64; ++i) {
                        const float want = i < n ? t.dct[c * M + m0 + i] : 0.f;
                        CHECK(std::memcmp(&want, &row[i], 4) == 0 || (want == 0.f && row[i] == 0.f), "lane %zu term %zu: %.9g, DCT table %.9g", lane, i, row[i], want);
                        if (i < n && (c & 1)) ++cover[c * (M / 2) + m0 + i];
                    }
                }
                for (size_t c = 1; c < Cc; c += 2)
                    for (size_t m = 0; m < M / 2; ++m) CHECK(cover[c * (M / 2) + m] == 1, "odd coefficient %zu term %zu covered %d times", c, m, cover[c * (M / 2) + m]);
            } else {
                for (size_t c = 0; c < Cc; ++c)
                    for (size_t m = 0; m < (M + 1) / 2; ++m) {
                        const float want = t.dct[c * M + m], got = f.tab[L::kCos + c * L::kCosPitch + m];
                        CHECK(std::memcmp(&want, &got, 4) == 0, "cos row %zu term %zu", c, m);
                    }
            }
        }
    }
    std::printf("%d configurations, %d valid; blocks built: fast512 %d, mfcc512w %d, mel512 %d, mfcc256 %d, mfcc1024 %d, mel1024 %d, mfcc2048 %d, mel2048 %d, mfcc4096 %d, mel4096 %d\n",
                cases, valid, built[0], built[1], built[2], built[3], built[4], built[5], built[6], built[7], built[8], built[9]);
    std::printf(g_fail ? "%d check(s) FAILED\n" : "all checks passed\n", g_fail);
    return g_fail ? 1 : 0;
}
