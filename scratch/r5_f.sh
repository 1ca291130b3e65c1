#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
o=gpurun_out/r5; mkdir -p $o
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider -k "async or sixel_batch or sixel_bytes" 2>&1 | tail -5
for m in "" "--sync-encode"; do
  echo "== bench $m"; timeout 300 python bench.py --no-cpu-baseline --no-extras --no-dropin $m 2>$o/bench_async.err | tail -1 > $o/bench_async$m.json
  python3 - $o/bench_async$m.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
print("ms/step",d["ms_per_step"],"value",d["value"],"scale",d["roofline"]["avg_launch_ms"],"frac",d["roofline"]["frac"],"stages",d["stages_ms"],"parity",(d.get("parity_check") or {}).get("ok"),"|",d["config"]["encode_call"][:40])
PY
done
tail -3 $o/bench_async.err
