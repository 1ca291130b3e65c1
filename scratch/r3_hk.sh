#!/bin/bash
scratch/run_logged.sh hk_pytest env TIMG_SKIP_CANARY=1 timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider -k "scale or horizontal or streaming or random or golden or config or full_size or width or triangle or bgra or composed"
tail -3 gpurun_out/r3/hk_pytest.log
for kind in photo alpha; do N=32 SW=7680 SH=4320 KIND=$kind timeout 200 python3 scratch/bench_scale.py 2>&1 | grep "^kernel"; done
N=64 SW=640 SH=480 DW=67 DH=50 KIND=alpha timeout 100 python3 scratch/bench_scale.py 2>&1 | grep "^kernel"
N=64 SW=1280 SH=960 DW=120 DH=90 KIND=alpha timeout 100 python3 scratch/bench_scale.py 2>&1 | grep "^kernel"
N=64 KIND=photo timeout 100 python3 scratch/bench_scale.py 2>&1 | grep "^kernel"
