#!/bin/bash
mkdir -p gpurun_out/r3
TIMG_SKIP_CANARY=1 timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -v -p no:cacheprovider -k "scale or horizontal or streaming or random or golden or config or full_size or width or triangle or bgra or composed" > gpurun_out/r3/hk3_pytest.txt 2>&1
echo "pytest rc=$?"
grep -v "^Extension modules" gpurun_out/r3/hk3_pytest.txt | tail -40 | cut -c1-300
dmesg 2>/dev/null | tail -20
