#!/bin/bash
# scratch/r5_b.sh -- decode alternatives in the vertical mix; the horizontal pass without its global load (timing only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
o=gpurun_out/r5; mkdir -p $o
echo "== vmix" ; timeout 120 scratch/ubench/vmix.bin 2>&1 | tee $o/vmix2.txt
echo "== bench_libs photo"; LIBS=main,r4,r6,hx,hx4,hx6 timeout 300 python3 scratch/bench_libs.py 2>&1 | grep -v "^$" | tee $o/bench_libs_noac.txt | tail -12
