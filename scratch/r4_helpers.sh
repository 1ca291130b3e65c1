#!/bin/bash
# scratch/r4_helpers.sh -- the diffusion's helper waves: pieces (columns) and naps (x 128 clocks) by TIMG_HIP_DITHER_HELPERS,
# metric configuration (DitherKernel's average duration under rocprofv3) and c2 (one frame, sixteen parts: ms per step)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
mkdir -p gpurun_out/r4
log=gpurun_out/r4/dither_helpers.txt; : > $log
for h in ${METRIC_H:-default 32,32 8,8}; do
  out=gpurun_out/r4/hlp; rm -rf "$out"; mkdir -p "$out"
  e=; [ $h != default ] && e="TIMG_HIP_DITHER_HELPERS=$h"
  env $e timeout -k 5 60 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o prof -- python bench.py --steps 4 --warmup 1 --no-dropin --no-parity --no-cpu-baseline --no-extras > "$out/log.txt" 2>&1
  f=$(find "$out" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$h" <<'PY' | tee -a $log
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "DitherKernel" in r["Name"]:
        print("metric helpers %-8s DitherKernel avg_us %9.1f" % (sys.argv[2], float(r["AverageNs"]) / 1e3))
PY
  rm -rf "$out"
done
for h in ${C2_H:-default 2,2 4,4 8,4}; do
  e=; [ $h != default ] && e="TIMG_HIP_DITHER_HELPERS=$h"
  env $e timeout -k 5 60 python bench.py --config c2 --steps 12 --warmup 4 --no-dropin --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c2 helpers %-8s ms_per_step %.3f encode %.3f parity %s' % ('$h', d['ms_per_step'], d['stages_ms']['encode'], d.get('parity_check',{}).get('ok')))" | tee -a $log
done
