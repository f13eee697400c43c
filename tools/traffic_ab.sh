#!/bin/bash
# On the GPU box: FETCH_SIZE / WRITE_SIZE of the dominant kernel for ab/lib_<name>.so builds (each counter in its own pass).
# usage: tools/traffic_ab.sh "name1 name2@ENV=val ..." [bench args]
V=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in $V; do
  n=${v%%@*}; e=""; [ "$v" != "$n" ] && e=${v#*@}
  for c in FETCH_SIZE WRITE_SIZE; do
    O=$R/gpurun_out/tab_${n}_${e//[^A-Za-z0-9]/_}_$c; rm -rf $O
    env $e SS_LIB_PATH=$R/ab/lib_$n.so rocprofv3 --pmc $c --output-format csv -d $O -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" > $O.log 2>&1
    python3 - $O $c "$v" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == sys.argv[2]:
            d[r["Kernel_Name"].split("(")[0][-60:]].append(float(r["Counter_Value"]))
for k, v in d.items():
    if len(v) >= 20:
        print("traffic", sys.argv[3], sys.argv[2], k, "KiB/launch %.1f" % (sum(v) / len(v)))
PY
  done
done
