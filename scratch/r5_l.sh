#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
o=gpurun_out/r5; mkdir -p $o
for cap in 8 12 16; do
  echo "== cap $cap"
  TIMG_HIP_TWIN_BATCH_CAP=$cap timeout 300 tests/twins/build/twin_bench --config metric,c4 --paths gpu,host --queue 4 --queue 17 --queue 33 --queue 64 --queue 129 --repeat 3 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-7s %-5s q%-4d %8.1f Mpx/s  %.3f ms/frame' % (d['config'], d['path'], d['queue_len'], d['mpx_per_s'], d['ms_per_frame']))" | tee $o/twin_cap_$cap.txt
done
