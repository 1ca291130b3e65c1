#!/bin/bash
# usage: scratch/pmc_scale.sh <name> "<counters>" [env...]
name=$1; shift; ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$name; rm -rf "$out"; mkdir -p "$out"
env "$@" rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out" -o pmc -- python scratch/bench_scale.py > "$out/log.txt" 2>&1
grep kernel "$out/log.txt" | head -3
f=$(find "$out" -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "ScaleStream" not in n: continue
    k = n.split("ScaleStream")[1][:22]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
rm -f "$out"/*kernel_trace.csv
