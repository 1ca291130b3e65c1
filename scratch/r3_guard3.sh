#!/bin/bash
for m in start end4; do echo "== $m"; TIMG_HIP_GUARD=$m timeout 300 python3 -X faulthandler scratch/r3_guard_stress.py 300 2>&1 | tail -4; done
bash scratch/r3_guard.sh
