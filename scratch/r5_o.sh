#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
o=gpurun_out/r5; mkdir -p $o
echo "== alpha frames (premultiplied set)"; KIND=alpha LIBS=main,pd4 timeout 300 python3 scratch/bench_libs.py 2>&1 | grep -v "^$\|amdgpu" | tee $o/bench_libs_alpha.txt | tail -5
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -p no:cacheprovider -k "not sixel and not gfx and not block and not png" 2>&1 | tail -4
