#!/bin/bash
# scratch/r6_a.sh -- the matrix scale kernel with typed buffer loads (-DTIMG_M_TYPED=1) beside the committed one: bytes and time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
{
LIBS=${LIBS:-main,typed} ROUNDS=9 timeout 300 python scratch/bench_libs.py 2>&1 | tail -12
} > "$out/typed_kernel.txt" 2>&1
cat "$out/typed_kernel.txt"
