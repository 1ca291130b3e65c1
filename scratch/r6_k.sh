#!/bin/bash
# scratch/r6_k.sh -- byte counts straight into pinned words (no memset / copy launches): parity, then the default bench line
# with and without (TIMG_HIP_SIXEL_COPY_LENGTHS=1), interleaved
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sixel or fused" 2>&1 | tail -3
for i in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export TIMG_HIP_SIXEL_COPY_LENGTHS=1; else unset TIMG_HIP_SIXEL_COPY_LENGTHS; fi
    python bench.py --steps 40 --warmup 5 --no-dropin --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('copy_lengths=$v', 'ms_per_step', d['ms_per_step'], 'value', d['value'], 'parity', d.get('parity_check'))"
  done
done | tee "$out/direct_lengths.txt"
