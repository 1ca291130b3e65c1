#!/bin/bash
# scratch/r6_m.sh <tag>... -- DitherKernel's average duration: main library and variant libraries, interleaved twice
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
for rep in 1; do
for tag in main "$@"; do
  d=$out/prof_v; rm -rf "$d"; mkdir -p "$d"
  lib=X=1; [ $tag != main ] && lib="TIMG_HIP_LIB=$GRAFT_REPO_ROOT/timg_amd/libtimg_hip_$tag.so"
  env $lib timeout -k 5 ${VT:-90} rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o prof -- python bench.py --steps 8 --warmup 2 --no-dropin --no-parity --no-cpu-baseline --no-extras > "$d/log.txt" 2>&1
  f=$(find "$d" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$tag" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "DitherKernel" in r["Name"]:
        print("%-12s DitherKernel avg_us %9.1f" % (sys.argv[2], float(r["AverageNs"]) / 1e3))
PY
  tail -3 "$d/log.txt" | cut -c1-200; rm -rf "$d"
done
done
