#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
TIMG_SKIP_CANARY=1 timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "fused or wide_upscales" 2>&1 | tail -5
for p in 1 2 4; do
  timeout 600 python3 bench.py --pieces $p --no-cpu-baseline --no-extras --no-dropin --steps 20 --warmup 5 > $O/bench_p$p.json 2> $O/bench_p$p.err; echo "pieces $p rc=$?"
  python3 - $p <<'PY'
import json,sys
p=sys.argv[1]
d=json.loads([l for l in open(f'gpurun_out/r3/bench_p{p}.json') if l.startswith('{')][-1])
print('  value',d['value'],'ms/step',d['ms_per_step'],'stages',d['stages_ms'],'roofline frac',d['roofline']['frac'],'launch ms',d['roofline']['avg_launch_ms'],'parity',d['parity_check']['ok'])
PY
done
GPU_MAX_HW_QUEUES=4 timeout 600 python3 bench.py --pieces 4 --no-cpu-baseline --no-extras --no-dropin --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('pieces 4 with 4 HW queues: ms/step',d['ms_per_step'],d['stages_ms'])"
