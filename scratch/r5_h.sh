#!/bin/bash
# scratch/r5_h.sh -- block canvas after the DPP scans: parity, then rocprofv3 per-kernel times of --mode quarter / half (config c3 shape and metric shape)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
o=gpurun_out/r5; mkdir -p $o
timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -p no:cacheprovider -k "block" 2>&1 | tail -4
: > $o/bench_modes.txt
for spec in "c3:quarter" "c3:half" "metric:quarter"; do
  cfg=${spec%%:*}; mode=${spec#*:}
  echo "== python bench.py --config $cfg --mode $mode --no-cpu-baseline --no-extras --no-dropin (rocprofv3 --kernel-trace --stats)" >> $o/bench_modes.txt
  bash profiles/prof.sh r5/prof_${cfg}_$mode --config $cfg --mode $mode --no-extras --no-dropin >> $o/bench_modes.txt 2>&1
done
cat $o/bench_modes.txt | cut -c1-220
