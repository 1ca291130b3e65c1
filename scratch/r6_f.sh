#!/bin/bash
# scratch/r6_f.sh -- held-batch cap x sequencer queue, after the sources moved to loader contexts
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
E=tests/twins/build/twin_bench
for cap in 4 8 16 32 64; do
  for q in 17 33 65 129; do
    r=$(TIMG_HIP_TWIN_BATCH_CAP=$cap $E --config metric --paths gpu --repeat 4 --queue $q 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Gpx/s %.2f ms' % (d['mpx_per_s']/1e3, d['seconds']*1e3))")
    echo "metric cap $cap queue $q: $r"
  done
  r=$(TIMG_HIP_TWIN_BATCH_CAP=$cap $E --config c4 --paths gpu --repeat 3 --queue 64 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Gpx/s %.2f ms' % (d['mpx_per_s']/1e3, d['seconds']*1e3))")
  echo "c4 cap $cap queue 64: $r"
done
