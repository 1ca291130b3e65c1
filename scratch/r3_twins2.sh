#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
timeout 1500 tests/twins/build/twin_bench --repeat 3 > $O/twin_bench.txt 2> $O/twin_bench.err; echo "twin_bench rc=$?"; python3 - <<'PY'
import json
for ln in open('gpurun_out/r3/twin_bench.txt'):
    if ln.startswith('{'):
        x=json.loads(ln); print(x['config'],x['path'],'q',x['queue_len'],'thr',x['loader_threads'],'frames',x['frames'],'ms/frame %.3f'%x['ms_per_frame'],'Mpx/s %.0f'%x['mpx_per_s'])
PY
tail -3 $O/twin_bench.err
