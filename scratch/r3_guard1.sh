#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
m=${1:-start}
TIMG_SKIP_CANARY=1 TIMG_HIP_GUARD=$m timeout 600 python3 -X faulthandler -m pytest tests/test_gpu_parity.py -k test_scale_bit_exact -m gpu -x -v -p no:cacheprovider > $O/guard1_$m.log 2>&1
echo "$m rc=$?"; grep -n "Memory access\|GUARD\|line .* in test_\|PASSED\|FAILED" $O/guard1_$m.log | tail -8
