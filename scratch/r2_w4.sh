#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r2w4; mkdir -p "$out"; rm -f $out/time.txt
timeout 200 python scratch/dbg_mfma.py 2>&1 | grep "bad px" | sort | uniq -c | tee "$out/dbg.txt"
for cfg in "A=0" "TIMG_HIP_M_WAVES=3" "TIMG_HIP_BAND_ROWS=30" "TIMG_HIP_BAND_ROWS=75" "TIMG_HIP_BAND_ROWS=90" "TIMG_HIP_BAND_ROWS=150"; do
    echo "== $cfg" | tee -a "$out/time.txt"
    env $cfg N=64 KIND=photo timeout 120 python scratch/bench_scale.py 2>&1 | grep "^kernel\|rror" | tee -a "$out/time.txt"
done
