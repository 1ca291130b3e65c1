#!/bin/bash
# scratch/r3_dither_parts.sh -- DitherKernel time (rocprofv3 kernel stats of the default bench step) by parts per frame
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
for parts in ${PARTS:-1 2 3 4}; do
  out=gpurun_out/r3/dparts_$parts; rm -rf "$out"; mkdir -p "$out"
  TIMG_HIP_DITHER_PARTS=$parts timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o prof -- python bench.py --steps 8 --no-dropin --no-parity --no-cpu-baseline --no-extras > "$out/log.txt" 2>&1
  f=$(find "$out" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" $parts "$out/log.txt" <<'PY'
import csv, sys, json
ms = None
for l in open(sys.argv[3]):
    if l.startswith("{"): ms = json.loads(l)["ms_per_step"]
for r in csv.DictReader(open(sys.argv[1])):
    if "DitherKernel" in r["Name"]:
        print("parts", sys.argv[2], "DitherKernel calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3), "min_us %.1f" % (float(r["MinNs"]) / 1e3), "ms_per_step", ms)
PY
  find "$out" -name '*kernel_trace.csv' -delete
done
