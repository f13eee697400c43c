// Host-side check of the 512-point kernel's mel tap layout (no GPU): prints, per slot, the lanes' first P bins and the LDS
// cycles one ds_read_b32 group of tap reads takes (1 = conflict-free) for the default bank and a few other shapes.
//   g++ -O2 -std=c++17 -Iinclude -Imfcc-rust_amd/csrc tools/hosttest/test_tap_banks.cpp mfcc-rust_amd/csrc/ss_host.cpp -o /tmp/tb && /tmp/tb
#include "ss_internal.h"

#include <cstdio>
#include <map>
#include <set>

static int group_cycles(const int32_t *start, int slot, int prow_pitch)
{
    // lanes (f, j), f = 0, 1: address = f * prow_pitch + start[slot * 16 + j] (+ tap index, the same for every lane)
    std::map<int, std::set<int>> banks;
    for (int f = 0; f < 2; ++f)
        for (int j = 0; j < 16; ++j) {
            const int a = f * prow_pitch + start[slot * 16 + j];
            banks[a & 31].insert(a);
        }
    size_t worst = 0;
    for (auto &b : banks) worst = std::max(worst, b.second.size());
    return static_cast<int>(worst);
}

// ds_read_b128 tap reads: four groups of 16 lanes; a group takes as many LDS cycles as its busiest float4 bank slot has
// distinct addresses.  Returns the cycles of all groups of the slot.
static int b128_cycles(const int32_t *start, int slot, int lanes)
{
    static const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int total = 0;
    for (int g = 0; g < lanes / 16; ++g) {
        std::map<int, std::set<int>> slots;
        for (int k = 0; k < 16; ++k) {
            const int lane = grp[g & 1][k] + 32 * (g / 2), a4 = start[slot * lanes + lane] / 4;
            slots[a4 & 15].insert(a4);
        }
        int worst = 0;
        for (auto &b : slots) worst = std::max<int>(worst, static_cast<int>(b.second.size()));
        total += worst;
    }
    return total;  // LDS cycles of one tap chunk of the slot; lanes / 16 when conflict-free
}

int main()
{
    struct Case { const char *name; uint32_t sr, filters; float len; };
    const Case cases[] = {{"default 16 kHz, 40 filters", 16000, 40, 0.02f}, {"16 kHz, 26 filters", 16000, 26, 0.02f},
                          {"8 kHz, 32 filters", 8000, 32, 0.02f}, {"22.05 kHz, 48 filters", 22050, 48, 0.02f}};
    int rc = 0;
    for (const Case &c : cases) {
        ss_params p;
        ss_params_default(&p, c.sr);
        p.num_filters = c.filters;
        p.frame_length = c.len;
        ss::HostTables t;
        if (ss::build_tables(p, t) != 0) { std::printf("%s: build_tables failed\n", c.name); rc = 1; continue; }
        ss::Fast512Tables f;
        ss::build_fast512(t, f);
        if (!f.ok) { std::printf("%s: no fast512 table\n", c.name); continue; }
        const int32_t *start = reinterpret_cast<const int32_t *>(f.tab.data() + ss::fast512_layout::kStart);
        std::printf("%s: q4 = %d %d %d, fullp %d\n", c.name, f.q4[0], f.q4[1], f.q4[2], f.fullp);
        for (int s = 0; s < 3; ++s) {
            if (!f.q4[s]) continue;
            std::printf("  slot %d first bins:", s);
            for (int j = 0; j < 16; ++j) std::printf(" %d", start[s * 16 + j]);
            const int cyc = group_cycles(start, s, f.fullp ? 576 : 144);
            std::printf("   -> %d LDS cycle(s) per tap group\n", cyc);
            if (!f.fullp && cyc != 1 && c.filters == 40) rc = 1;
        }
    }
    {   // cfg5: 44.1 kHz, fft 4096, 256 filters (one P row per wave, 64 lanes)
        ss_params p;
        ss_params_default(&p, 44100);
        p.fft_points = 4096; p.frame_length = 4096.f / 44100.f; p.frame_stride = 1024.f / 44100.f; p.num_cepstral = 40; p.num_filters = 256;
        ss::HostTables t;
        ss::Mfcc4096Tables f;
        if (ss::build_tables(p, t) == 0) ss::build_mfcc4096(t, f);
        if (!f.ok) { std::printf("cfg5: no table\n"); rc = 1; }
        else {
            // (this layout packs the filter index into the upper half of each word)
            int32_t start[256];
            for (int q = 0; q < 256; ++q) start[q] = reinterpret_cast<const int32_t *>(f.tab.data() + ss::mfcc4096_layout::kStart)[q] & 0xffff;
            for (int s = 0; s < 4; ++s) {
                const int cyc = b128_cycles(start, s, 64);
                std::printf("cfg5 slot %d (q4 %d): %d LDS cycles per tap chunk (4 conflict-free)\n", s, f.q4[s], cyc);
                if (cyc > 6) rc = 1;
            }
        }
    }
    {   // cfg3: 16 kHz, fft 2048, 128 filters (32 lanes per P row)
        ss_params p;
        ss_params_default(&p, 16000);
        p.fft_points = 2048; p.frame_length = 0.032f; p.frame_stride = 0.032f; p.num_filters = 128; p.high_frequency = 8000.f;
        ss::HostTables t;
        ss::Mel2048Tables f;
        if (ss::build_tables(p, t) == 0) ss::build_mel2048(t, f);
        if (!f.ok) { std::printf("cfg3: no table\n"); rc = 1; }
        else {
            const int32_t *start = reinterpret_cast<const int32_t *>(f.tab.data() + ss::mel2048_layout::kStart);
            for (int s = 0; s < 4; ++s) {
                const int cyc = b128_cycles(start, s, 32);
                std::printf("cfg3 slot %d (q4 %d): %d LDS cycles per tap chunk (2 conflict-free)\n", s, f.q4[s], cyc);
                if (cyc > 4) rc = 1;
            }
        }
    }
    return rc;
}
