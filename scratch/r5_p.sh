#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
o=gpurun_out/r5; mkdir -p $o
timeout 300 tests/twins/build/twin_check timggrid 2>&1 | tail -2; timeout 300 tests/twins/build/twin_check grid 2>&1 | tail -2
timeout -k 5 500 tests/twins/build/twin_bench --repeat 3 --cpu-frames 64 > $o/twin_bench.txt 2> $o/twin_bench.err
python3 - <<'PY'
import json
for l in open("gpurun_out/r5/twin_bench.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        print("%-7s %-5s q%-4d frames %4d threads %3d  %8.1f Mpx/s  %.3f ms/frame" % (d["config"], d["path"], d["queue_len"], d["frames"], d["loader_threads"], d["mpx_per_s"], d["ms_per_frame"]))
PY
