#!/bin/bash
# scratch/r5_stress.sh -- random geometries against the oracle / the generic kernel for a couple of minutes (round 5's last binary)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
o=gpurun_out/r5; mkdir -p $o
timeout 200 python3 scratch/stress.py 2>&1 | tail -4 | tee $o/stress.txt
timeout 150 python3 scratch/scale_stress.py 2>&1 | tail -3 | tee -a $o/stress.txt
timeout 150 python3 scratch/sixel_stress.py 60 2>&1 | tail -3 | tee -a $o/stress.txt
