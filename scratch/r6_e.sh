#!/bin/bash
# scratch/r6_e.sh -- the drop-in path after the sources moved to loader contexts: timeline, default queues, twin / binary tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
E=tests/twins/build/twin_bench
{ TWIN_BENCH_TIMELINE=1 $E --config metric --paths gpu --repeat 3 --queue 33 2>&1 | tail -4; } > "$out/twin_timeline_after.txt"
$E --config c2,c3,c4,metric --repeat 3 --cpu-frames 64 2>/dev/null | grep '^{' > "$out/twin_bench.txt"
python3 - <<'PY'
import json
for l in open("gpurun_out/r6/twin_bench.txt"):
    d = json.loads(l); print(d["config"], d["path"], "queue", d["queue_len"], "%.1f Gpx/s" % (d["mpx_per_s"] / 1e3), "%.3f ms/frame" % d["ms_per_frame"])
PY
timeout 1200 python -m pytest tests/test_twins.py tests/test_timg_binary.py -x -q -m gpu 2>&1 | tail -3
