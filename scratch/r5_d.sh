#!/bin/bash
# scratch/r5_d.sh -- the patched timg binary on the device + the twin checks touched this round
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
o=gpurun_out/r5; mkdir -p $o
timeout 600 python3 -m pytest tests/test_timg_binary.py -x -q -p no:cacheprovider 2>&1 | tail -15
echo "== twin_check hostpath scaler"; timeout 300 tests/twins/build/twin_check scaler 2>&1 | tail -3; timeout 300 tests/twins/build/twin_check hostpath 2>&1 | tail -3
