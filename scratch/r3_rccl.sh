#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
TIMG_SKIP_CANARY=1 timeout 600 python3 -m pytest tests/test_zz_rccl.py -m gpu -v -s -p no:cacheprovider 2>&1 | tail -15
echo "== without HSA_ENABLE_IPC_MODE_LEGACY"; TIMG_SKIP_CANARY=1 env -u HSA_ENABLE_IPC_MODE_LEGACY timeout 600 python3 -m pytest tests/test_zz_rccl.py -m gpu -q -s -p no:cacheprovider 2>&1 | tail -5
