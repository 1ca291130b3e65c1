#!/bin/bash
for lib in libtimg_hip.so libtimg_hip_p4.so libtimg_hip.so libtimg_hip_p4.so; do
  echo "== $lib"; TIMG_HIP_LIB=$PWD/timg_amd/$lib N=64 KIND=alpha timeout 120 python3 scratch/bench_scale.py 2>&1 | grep "^kernel" | cut -c1-140
done
TIMG_HIP_LIB=$PWD/timg_amd/libtimg_hip_p4.so TIMG_SKIP_CANARY=1 timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider -k "scale or blend or golden or fused or full_size or matrix or streaming or random or config" 2>&1 | tail -2
timeout 100 python3 scratch/scale_stress.py 30 77 2>&1 | grep -c MISMATCH
