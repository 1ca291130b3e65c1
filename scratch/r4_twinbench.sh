#!/bin/bash
# scratch/r4_twinbench.sh [twin_bench args] -- the drop-in path as src/timg.cc drives it (tests/twins/twin_bench)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
mkdir -p gpurun_out/r4
timeout -k 5 400 tests/twins/build/twin_bench "$@" > gpurun_out/r4/twin_bench.txt 2> gpurun_out/r4/twin_bench.err
python3 - <<'PY'
import json
for l in open("gpurun_out/r4/twin_bench.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        print("%-7s %-5s q%-4d frames %4d threads %3d  %8.1f Mpx/s  %.3f ms/frame" % (d["config"], d["path"], d["queue_len"], d["frames"], d["loader_threads"], d["mpx_per_s"], d["ms_per_frame"]))
PY
tail -3 gpurun_out/r4/twin_bench.err
