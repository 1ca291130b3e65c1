#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
TIMG_SKIP_CANARY=1 timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider -k "scale or blend or golden or fused or full_size or matrix or streaming or random or autocrop" 2>&1 | tail -4
for i in 1 2; do
timeout 600 python3 bench.py --no-cpu-baseline --no-extras --no-dropin --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms/step',d['ms_per_step'],d['stages_ms'],'frac',d['roofline']['frac'],'launch',d['roofline']['avg_launch_ms'],'parity',d['parity_check']['ok'])"
done
