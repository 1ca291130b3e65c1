#!/bin/bash
# scratch/r6_h2.sh -- the two-columns-per-lane-pair horizontal-first kernel: parity subset, then time beside the old kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "horizontal_first or config5 or random_geometries or any_source_width or transparent_pixels or fallback_chain or scale_bit_exact or config1" 2>&1 | tail -8
N=${N:-32} SW=7680 SH=4320 KIND=alpha VARIANTS="h2:;old:TIMG_HIP_H2=0" ROUNDS=5 REPS=5 timeout 600 python scratch/bench_scale.py 2>&1 | grep "^kernel\|rror\|{"
N=${N:-32} SW=7680 SH=4320 KIND=photo VARIANTS="h2:;old:TIMG_HIP_H2=0" ROUNDS=5 REPS=5 timeout 600 python scratch/bench_scale.py 2>&1 | grep "^kernel\|rror"
} > "$out/h2_kernel.txt" 2>&1
cat "$out/h2_kernel.txt"
