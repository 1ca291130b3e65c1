#!/bin/bash
scratch/run_logged.sh sixel_pytest env TIMG_SKIP_CANARY=1 timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "sixel" --durations=5
tail -12 gpurun_out/r3/sixel_pytest.log
