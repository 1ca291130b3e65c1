#!/bin/bash
# scratch/r6_h2sweep.sh -- band heights x builds of the two-column kernel, 64 8K S-alpha frames (BASELINE config 5's chunk)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
for tag in ${TAGS:-h2w2 h2w3d2}; do
  echo "== $tag"
  TIMG_HIP_LIB=$PWD/timg_amd/libtimg_hip_$tag.so N=64 SW=7680 SH=4320 KIND=alpha ROUNDS=3 REPS=3 WARM_S=0.2 \
  VARIANTS="old90:TIMG_HIP_H2=0;b90:TIMG_HIP_BAND_ROWS=90;b75:TIMG_HIP_BAND_ROWS=75;b64:TIMG_HIP_BAND_ROWS=64;b56:TIMG_HIP_BAND_ROWS=56;b50:TIMG_HIP_BAND_ROWS=50;b45:TIMG_HIP_BAND_ROWS=45;b38:TIMG_HIP_BAND_ROWS=38;b30:TIMG_HIP_BAND_ROWS=30" \
  timeout 600 python scratch/bench_scale.py 2>&1 | grep "^kernel\|rror"
done > "$out/h2_sweep.txt" 2>&1
cat "$out/h2_sweep.txt"
