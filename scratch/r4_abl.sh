#!/bin/bash
# scratch/r4_abl.sh <tag>... -- DitherKernel's average duration (rocprofv3 kernel stats of the default bench step) for the
# main library and each libtimg_hip_<tag>.so (timing experiments: scratch/build_variant.sh aN sixel_canvas.hip -DTIMG_DITHER_ABL=N)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
mkdir -p gpurun_out/r4
: > gpurun_out/r4/dither_ablation.txt
for tag in main "$@"; do
  out=gpurun_out/r4/abl_$tag; rm -rf "$out"; mkdir -p "$out"
  lib=; [ $tag != main ] && lib="TIMG_HIP_LIB=$GRAFT_REPO_ROOT/timg_amd/libtimg_hip_$tag.so"
  env $lib timeout -k 5 60 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o prof -- python bench.py --steps 4 --warmup 1 --no-dropin --no-parity --no-cpu-baseline --no-extras > "$out/log.txt" 2>&1
  f=$(find "$out" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$tag" <<'PY' | tee -a gpurun_out/r4/dither_ablation.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "DitherKernel" in r["Name"]:
        print("%-6s DitherKernel calls %4s avg_us %9.1f" % (sys.argv[2], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf "$out"
done
