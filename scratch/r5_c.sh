#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
o=gpurun_out/r5; mkdir -p $o
echo "== bench_libs photo"; LIBS=${LIBS:-main,s0,s4,s6,r6} timeout 300 python3 scratch/bench_libs.py 2>&1 | grep -v "^$" | tee $o/bench_libs_${TAG:-c}.txt | tail -14
