#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
o=gpurun_out/r5; mkdir -p $o
timeout 300 tests/twins/build/twin_check sixel 2>&1 | tail -2; timeout 300 tests/twins/build/twin_check timggrid 2>&1 | tail -2; timeout 300 tests/twins/build/twin_check sixelgrid 2>&1 | tail -2
timeout 300 tests/twins/build/twin_bench --config metric,c4 --paths gpu,host --queue 4 --queue 17 --queue 33 --queue 64 --queue 129 --repeat 3 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-7s %-5s q%-4d %8.1f Mpx/s  %.3f ms/frame' % (d['config'], d['path'], d['queue_len'], d['mpx_per_s'], d['ms_per_frame']))" | tee $o/twin_exact_buffers.txt
