#!/bin/bash
# scratch/r6_g.sh <tag>... -- DitherKernel's average duration for the main library and ablation variants, all with one pixel a
# request (TIMG_HIP_DITHER_PIX=1: the 512 ablation -- pixels and cells through LDS, as helper waves would stage them -- is
# written for that form)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
mkdir -p gpurun_out/r6
: > gpurun_out/r6/dither_staged_ablation.txt
for tag in main "$@" main; do
  out=gpurun_out/r6/abl_$tag; rm -rf "$out"; mkdir -p "$out"
  lib=; [ $tag != main ] && lib="TIMG_HIP_LIB=$GRAFT_REPO_ROOT/timg_amd/libtimg_hip_$tag.so"
  env $lib TIMG_HIP_DITHER_PIX=1 timeout -k 5 90 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o prof -- python bench.py --steps 6 --warmup 2 --no-dropin --no-parity --no-cpu-baseline --no-extras > "$out/log.txt" 2>&1
  f=$(find "$out" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$tag" <<'PY' | tee -a gpurun_out/r6/dither_staged_ablation.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "DitherKernel" in r["Name"]:
        print("%-6s DitherKernel calls %4s avg_us %9.1f" % (sys.argv[2], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  tail -2 "$out/log.txt" | cut -c1-300
  rm -rf "$out"
done
