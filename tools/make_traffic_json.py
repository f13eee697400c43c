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
# shader clock the profiled box held under the load (SS_PROFILE_CLOCK_GHZ: bench.py's clock_ghz_measured of the same box;
# default: the 1.974 GHz of profiles/r02/wave_timeline_cfg2.txt)
CLOCK_GHZ = float(os.environ.get("SS_PROFILE_CLOCK_GHZ", "1.974"))
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
        "profiled": os.environ.get("SS_PROFILE_TAG", "round 2"),
    }
    # SQ counters.  SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles summed over the waves: their ratios are
    # shares of the waves' lifetime.  SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT count LDS-array cycles summed over the CUs.
    if d.get("SQ_WAVE_CYCLES"):
        out[wl]["valu_insts_per_launch"] = d.get("SQ_INSTS_VALU")
        out[wl]["lds_insts_per_launch"] = d.get("SQ_INSTS_LDS")
        out[wl]["wave_time_shares"] = {k: d.get(k, 0.0) / d["SQ_WAVE_CYCLES"] for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS")}
        # VALU issue floor measured in tools/ubench/valu_issue.hip: 2.14 cycles per instruction per SIMD with >= 2 waves.
        # Clock: the shader clock during a cfg2 launch, measured with s_memtime against s_memrealtime (tools/dbg_times.py):
        # 1.97 GHz (1.90-2.07) -- the part does not hold 2.4 GHz under this load.
        if d.get("avg_ns"):
            out[wl]["clock_ghz_assumed"] = CLOCK_GHZ
            out[wl]["valu_floor_frac"] = d.get("SQ_INSTS_VALU", 0.0) / 1024.0 * 2.14 / (d["avg_ns"] * CLOCK_GHZ)
    if d.get("SQ_LDS_IDX_ACTIVE"):
        out[wl]["lds_bank_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
        if d.get("avg_ns"):
            out[wl]["lds_busy_frac"] = d["SQ_LDS_IDX_ACTIVE"] / 256.0 / (d["avg_ns"] * CLOCK_GHZ)
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1))
