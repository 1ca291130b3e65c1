#!/bin/bash
# scratch/r3_final.sh -- the round's evidence in one GPU call: profiles/collect.sh, the BASELINE configs, the default line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
bash profiles/collect.sh r3 > gpurun_out/r3_collect.log 2>&1; tail -3 gpurun_out/r3_collect.log | cut -c1-400
out=gpurun_out/r3/bench_configs.txt; : > $out
for c in c2 c3 c4 c5; do
  echo "== python bench.py --config $c" >> $out
  timeout 400 python bench.py --config $c 2>>gpurun_out/r3/bench_configs.err | tail -1 >> $out
done
echo "== python bench.py" >> $out
timeout 300 python bench.py 2>>gpurun_out/r3/bench_configs.err | tail -1 >> $out
cut -c1-330 $out
