#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
scratch/run_logged.sh sixel_pytest env TIMG_SKIP_CANARY=1 timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider -k "sixel or config or golden"
tail -2 gpurun_out/r3/sixel_pytest.log
timeout 300 python bench.py --no-dropin > gpurun_out/r3/bench_dither.json 2>gpurun_out/r3/bench_dither.err
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3/bench_dither.json") if l.startswith("{")][-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "parity", d.get("parity_check", {}).get("ok") if isinstance(d.get("parity_check"), dict) else d.get("parity_check"))
for k in d:
    if "batched" in k or "d2h" in k or "alpha" in k: print(k, json.dumps(d[k])[:300])
PY
