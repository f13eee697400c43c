#!/usr/bin/env python3
"""profiles/pmc_traffic.json from tools/profile.sh summaries: HBM bytes per launch of the dominant kernel.

MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE come from the L2's fabric-side request
counters, in KiB; FETCH_SIZE and WRITE_SIZE need separate --pmc passes (TCC has 4 slots); on gfx950 FETCH_SIZE
reports exactly 1/2 of the bytes of a coalesced streaming read and must be doubled (checked here against
the known floor: every input sample has to be fetched at least once); WRITE_SIZE reads exact.
usage: tools/make_traffic_json.py <workload>=<gpurun_out/prof_x/summary.json> ...
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
out = json.load(open(out_path)) if os.path.exists(out_path) else {}
for arg in sys.argv[1:]:
    wl, path = arg.split("=", 1)
    summ = json.load(open(path))
    name, d = max(summ.items(), key=lambda kv: kv[1].get("avg_ns", 0) * kv[1].get("calls", 0))
    short = name.split("::")[-1].split("(")[0].replace(", ", ",")
    out[wl] = {
        "kernel_full": name,
        "fetch_size_kib": d.get("FETCH_SIZE"),
        "write_size_kib": d.get("WRITE_SIZE"),
        "hbm_bytes_per_launch": 2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024,
        "correction": "2 x FETCH_SIZE (gfx950 coalesced-read under-count) + WRITE_SIZE, KiB -> bytes",
        "avg_ns_profiled": d.get("avg_ns"),
    }
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1))
