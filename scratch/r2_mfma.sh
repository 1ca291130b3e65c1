#!/bin/bash
# scratch/r2_mfma.sh -- scale parity tests, then timings with and without the matrix-core path
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r2mfma; mkdir -p "$out"; rm -f $out/time.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "scale or blend or golden or stream or full_size or config or triangle or autocrop or width" 2>&1 | grep -E "passed|failed|Error" | tail -4 | tee "$out/pytest.txt"
for kind in photo alpha; do
  for cfg in "A=0" "TIMG_HIP_NO_MATRIX=1"; do
    echo "== kind=$kind $cfg" | tee -a "$out/time.txt"
    env $cfg N=64 KIND=$kind timeout 120 python scratch/bench_scale.py 2>&1 | grep "^kernel\|rror" | tee -a "$out/time.txt"
  done
done
