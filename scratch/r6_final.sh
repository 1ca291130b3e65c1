#!/bin/bash
# scratch/r6_final.sh -- the round's evidence in one GPU call: profiles/collect.sh (kernel stats of the default command), the
# BASELINE configs, the default line, twin_bench, then the counters of every configuration (each file carries the SHA-256 of the
# library it was collected with)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
mkdir -p gpurun_out/r6
timeout -k 5 400 bash profiles/collect.sh r6x > gpurun_out/r6_collect.log 2>&1; tail -3 gpurun_out/r6_collect.log | cut -c1-300
cp gpurun_out/r6x/summary.txt gpurun_out/r6x/kernel_stats.csv gpurun_out/r6x/hbm_traffic.json gpurun_out/r6/ 2>/dev/null
out=gpurun_out/r6/bench_configs.txt; : > $out
for c in c2 c3 c4 c5; do
  echo "== python bench.py --config $c" >> $out
  timeout -k 5 400 python bench.py --config $c 2>>gpurun_out/r6/bench_configs.err | tail -1 >> $out
done
echo "== python bench.py" >> $out
timeout -k 5 400 python bench.py 2>>gpurun_out/r6/bench_configs.err | tail -1 >> $out
cut -c1-260 $out
timeout -k 5 500 tests/twins/build/twin_bench --repeat 2 --cpu-frames 64 > gpurun_out/r6/twin_bench.txt 2> gpurun_out/r6/twin_bench.err
python3 - <<'PY'
import json
for l in open("gpurun_out/r6/twin_bench.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        print("%-7s %-5s q%-4d frames %4d threads %3d  %8.1f Mpx/s  %.3f ms/frame" % (d["config"], d["path"], d["queue_len"], d["frames"], d["loader_threads"], d["mpx_per_s"], d["ms_per_frame"]))
PY
bash profiles/collect_pmc.sh r6 2>&1 | cut -c1-400 | tail -8
