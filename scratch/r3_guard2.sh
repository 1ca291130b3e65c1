#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
TIMG_HIP_GUARD=start timeout 300 python3 -X faulthandler scratch/r3_guard_stress.py > $O/stress.log 2>&1; echo "stress rc=$?"; tail -3 $O/stress.log
TIMG_SKIP_CANARY=1 TIMG_HIP_GUARD=start AMD_LOG_LEVEL=3 timeout 600 python3 -X faulthandler -m pytest tests/test_gpu_parity.py -k test_scale_bit_exact -m gpu -x -v -p no:cacheprovider > $O/guard_amdlog.log 2>&1
echo "rc=$?"; grep -v "^:3:" $O/guard_amdlog.log | head -30; grep -n "Fatal Python" $O/guard_amdlog.log | head -2
L=$(grep -n "Fatal Python" $O/guard_amdlog.log | head -1 | cut -d: -f1); head -n $L $O/guard_amdlog.log | tail -n 60 > $O/guard_amdlog_tail.txt; rm $O/guard_amdlog.log
