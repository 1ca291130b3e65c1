#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
o=gpurun_out/r5; mkdir -p $o
echo "== c5 geometry: band rows"; N=64 SW=7680 SH=4320 KIND=alpha REPS=4 ROUNDS=3 VARIANTS="b45:TIMG_HIP_BAND_ROWS=45;b57:TIMG_HIP_BAND_ROWS=57;b75:TIMG_HIP_BAND_ROWS=75;b90:TIMG_HIP_BAND_ROWS=90;b113:TIMG_HIP_BAND_ROWS=113;b150:TIMG_HIP_BAND_ROWS=150;b225:TIMG_HIP_BAND_ROWS=225" timeout 300 python3 scratch/bench_scale.py 2>&1 | grep "^kernel\|rror" | tee $o/band_rows_c5.txt
