#!/bin/bash
# scratch/r5_t.sh -- the opaque horizontal pass with B as a packed tap pair: store / read variants against the committed build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r5; mkdir -p "$out"
{
LIBS=base,hba1,hba3,hba3r,hbb2,main ROUNDS=9 timeout 300 python scratch/bench_libs.py 2>&1 | tail -12
} > "$out/hpass_bpair.txt" 2>&1
cat "$out/hpass_bpair.txt"
