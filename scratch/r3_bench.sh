#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
TIMG_SKIP_CANARY=1 timeout 900 python3 -m pytest tests/test_zz_rccl.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3/bench_default.json') if l.startswith('{')][-1])
for k in ['value','ms_per_step','stages_ms','parity_check','rccl']: print(k, d.get(k))
print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'])
print('cpp_dropin', json.dumps(d.get('cpp_dropin'))[:1500])
PY
tail -3 $O/bench_default.err
