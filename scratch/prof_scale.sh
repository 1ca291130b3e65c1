#!/bin/bash
# usage: scratch/prof_scale.sh <name> [env...]  -- kernel-trace stats of scratch/bench_scale.py
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$name; rm -rf "$out"; mkdir -p "$out"
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o prof -- python scratch/bench_scale.py > "$out/log.txt" 2>&1
grep kernel "$out/log.txt"
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "timg_amd" not in n: continue
    print("%-60s calls %4s avg_us %10.1f" % (n.split("(anonymous namespace)::")[-1][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find "$out" -name '*kernel_trace.csv' -delete
