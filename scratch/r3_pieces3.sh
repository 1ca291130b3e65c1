#!/bin/bash
run() { timeout 300 python3 bench.py --pieces $1 --no-cpu-baseline --no-extras --no-dropin --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('pieces $1 band ${TIMG_HIP_BAND_ROWS:-45}: ms/step',d['ms_per_step'],d['stages_ms'],'frac',d['roofline']['frac'],'parity',d['parity_check']['ok'])"; }
run 1; run 2; run 3; run 4
TIMG_HIP_BAND_ROWS=65 run 2
TIMG_HIP_BAND_ROWS=57 run 2
