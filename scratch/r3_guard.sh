#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
for m in start end16 end4; do
  TIMG_SKIP_CANARY=1 TIMG_HIP_GUARD=$m timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -v -p no:cacheprovider > $O/guard_$m.log 2>&1
  echo "$m rc=$?" | tee -a $O/guard_$m.log
  grep -n "Memory access\|GUARD\|line .* in test_\|rc=" $O/guard_$m.log | head
  tail -n 3 $O/guard_$m.log
done
