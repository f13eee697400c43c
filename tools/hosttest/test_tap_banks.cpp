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
    return rc;
}
