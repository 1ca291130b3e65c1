#!/bin/bash
# tools/prof.sh <outdir-name> <bench args...>: rocprofv3 kernel trace + stats of bench.py on the
# GPU box (run through gpurun); writes CSVs under gpurun_out/<name>/ and prints the timg kernels.
set -e
ulimit -c 0
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$name
rm -rf "$out"; mkdir -p "$out"
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o prof -- \
    python bench.py --no-cpu-baseline "$@" > "$out/bench.log" 2>&1 || { tail -20 "$out/bench.log"; exit 1; }
tail -1 "$out/bench.log" | cut -c1-400
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-28s %6s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for r in rows:
    n = r["Name"]
    if "timg_amd" not in n: continue
    short = n.split("(anonymous namespace)::")[1].split("(")[0] if "(anonymous namespace)::" in n else n.split("(")[0]
    if "ScaleStreamKernel" in n: short = "ScaleStreamKernel" + n.split("ScaleStreamKernel")[1].split("(")[0]
    print("%-28s %6s %12.1f %12.1f %7.2f" % (short[:28], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                           float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
# keep only the small summaries
find "$out" -name '*kernel_trace.csv' -delete
