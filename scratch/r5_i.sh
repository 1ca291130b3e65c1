#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
o=gpurun_out/r5; mkdir -p $o
for lib in main s6; do
  [ $lib = main ] && L=timg_amd/libtimg_hip.so || L=timg_amd/libtimg_hip_$lib.so
  echo "== $lib: band rows"; TIMG_HIP_LIB=$GRAFT_REPO_ROOT/$L VARIANTS="b45:TIMG_HIP_BAND_ROWS=45;b57:TIMG_HIP_BAND_ROWS=57;b75:TIMG_HIP_BAND_ROWS=75;b90:TIMG_HIP_BAND_ROWS=90;b113:TIMG_HIP_BAND_ROWS=113;b150:TIMG_HIP_BAND_ROWS=150" timeout 200 python3 scratch/bench_scale.py 2>&1 | grep "^kernel" | tee -a $o/band_rows.txt
done
