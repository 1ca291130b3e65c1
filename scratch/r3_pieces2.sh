#!/bin/bash
run() { timeout 300 python3 bench.py --pieces $1 --no-cpu-baseline --no-extras --no-dropin --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('pieces $1 band ${TIMG_HIP_BAND_ROWS:-45}: ms/step',d['ms_per_step'],d['stages_ms'],'parity',d['parity_check']['ok'])"; }
run 1
for b in 65 57 75 90 113; do TIMG_HIP_BAND_ROWS=$b run 2; done
TIMG_HIP_BAND_ROWS=65 run 1
TIMG_HIP_BAND_ROWS=113 run 4
TIMG_HIP_BAND_ROWS=150 run 4
