#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
scratch/run_logged.sh sixel_pytest env TIMG_SKIP_CANARY=1 timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider -k "sixel or config or golden"
tail -2 gpurun_out/r3/sixel_pytest.log
TIMG_HIP_LIB=$GRAFT_REPO_ROOT/timg_amd/libtimg_hip_ct.so timeout 300 python3 scratch/cut_trace.py 2>&1 | grep "^cut:" | tail -26 | head -9
TIMG_HIP_LIB=$GRAFT_REPO_ROOT/timg_amd/libtimg_hip_ct.so timeout 300 python3 scratch/cut_trace.py 2>&1 | grep "^cut: rounds" | tail -1
