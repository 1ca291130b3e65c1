#!/bin/bash
# scratch/r6_l.sh -- the diffusion's helper waves (columns a piece, naps between polls) and parts a frame, after the step became faster
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
: > "$out/dither_helpers.txt"
run() {  # <label> <env...>
  label=$1; shift
  d=$out/prof_h; rm -rf "$d"; mkdir -p "$d"
  env "$@" timeout -k 5 90 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o prof -- python bench.py --steps 8 --warmup 2 --no-dropin --no-parity --no-cpu-baseline --no-extras > "$d/log.txt" 2>&1
  f=$(find "$d" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$label" <<'PY' | tee -a "$out/dither_helpers.txt"
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "DitherKernel" in r["Name"]:
        print("%-28s DitherKernel avg_us %9.1f" % (sys.argv[2], float(r["AverageNs"]) / 1e3))
PY
  rm -rf "$d"
}
run default X=1
for h in 1,1 2,2 4,2 2,4 4,4 8,4 8,8 16,8; do run "helpers $h" TIMG_HIP_DITHER_HELPERS=$h; done
for p in 2 3 4; do run "parts $p" TIMG_HIP_DITHER_PARTS=$p; done
run "trips 2" TIMG_HIP_DITHER_TRIPS=2
run "pix 1" TIMG_HIP_DITHER_PIX=1
run default X=1
