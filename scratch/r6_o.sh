#!/bin/bash
# scratch/r6_o.sh -- DitherKernel by frames a batch (does the pixel request's latency show: few frames stay in the L2)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
for n in 1 4 16 64 128; do
  d=$out/prof_f; rm -rf "$d"; mkdir -p "$d"
  timeout -k 5 90 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o prof -- python bench.py --frames $n --steps 8 --warmup 2 --no-dropin --no-parity --no-cpu-baseline --no-extras > "$d/log.txt" 2>&1
  f=$(find "$d" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$n" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+Kernel(?:<[^>]*>)?)", r["Name"])
    if m and ("Dither" in m.group(1) or "MedianCut" in m.group(1) or "BandNodes" in m.group(1)):
        print("frames %4s %-40s avg_us %8.1f" % (sys.argv[2], m.group(1), float(r["AverageNs"]) / 1e3))
PY
  rm -rf "$d"
done | tee "$out/dither_by_frames.txt"
