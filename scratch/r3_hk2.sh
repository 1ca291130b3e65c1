#!/bin/bash
mkdir -p gpurun_out/r3
TIMG_SKIP_CANARY=1 timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "composed" > gpurun_out/r3/hk2_pytest.txt 2>&1
grep -v "^Extension modules" gpurun_out/r3/hk2_pytest.txt | head -60
N=64 SW=640 SH=480 DW=600 DH=450 KIND=alpha TIMG_HIP_NO_MATRIX=1 timeout 100 python3 scratch/bench_scale.py 2>&1 | grep -v "^Extension" | tail -5
timeout 300 python3 bench.py --config c5 --no-dropin 2>&1 | grep -v "^Extension" | tail -3 | cut -c1-600
