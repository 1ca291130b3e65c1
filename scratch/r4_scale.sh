#!/bin/bash
# scratch/r4_scale.sh -- the scale parity tests, then one bench line per configuration (no CPU baseline / extras)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
mkdir -p gpurun_out/r4
TIMG_ROUND=r4 scratch/run_logged.sh scale_pytest env TIMG_SKIP_CANARY=1 timeout -k 5 400 python3 -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider -k "not sixel and not gfx and not block and not png"
tail -4 gpurun_out/r4/scale_pytest.log | cut -c1-300
out=gpurun_out/r4/bench_scale_configs.txt; : > $out
for c in ${CONFIGS:-metric c2 c3 c5}; do
  echo "== python bench.py --config $c --no-cpu-baseline --no-extras --no-dropin" >> $out
  timeout -k 5 200 python bench.py --config $c --no-cpu-baseline --no-extras --no-dropin 2>>gpurun_out/r4/bench_scale_configs.err | tail -1 >> $out
done
python3 - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["config"]["workload"][:40], "ms/step", d["ms_per_step"], "value", d["value"], "scale ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"],
              "traffic_x", d["roofline"].get("traffic_over_algorithmic"), "stages", d["stages_ms"], "parity", (d.get("parity_check") or {}).get("ok"))
    else:
        print(l.strip())
PY
