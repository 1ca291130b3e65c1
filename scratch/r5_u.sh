#!/bin/bash
# scratch/r5_u.sh -- do the workgroups of a CU complete their rows in phase?  a start-up stagger by the tile (s_sleep units of 64 clocks)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r5; mkdir -p "$out"
{
LIBS=main,stg12,stg24,stg48,stg96 ROUNDS=9 timeout 300 python scratch/bench_libs.py 2>&1 | tail -10
} > "$out/stagger.txt" 2>&1
cat "$out/stagger.txt"
