#!/bin/bash
# scratch/r4_twins.sh -- the twins' GPU tests (tests/test_twins.py, tests/test_abi.py) and the sixel parity subset
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
mkdir -p gpurun_out/r4
TIMG_ROUND=r4 scratch/run_logged.sh twins_pytest env TIMG_SKIP_CANARY=1 timeout -k 5 500 python3 -X faulthandler -m pytest tests/test_twins.py tests/test_abi.py -m gpu -q -p no:cacheprovider --tb=short
tail -25 gpurun_out/r4/twins_pytest.log | cut -c1-400
TIMG_ROUND=r4 scratch/run_logged.sh sixel_pytest env TIMG_SKIP_CANARY=1 timeout -k 5 240 python3 -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider -k "sixel or config or golden"
tail -4 gpurun_out/r4/sixel_pytest.log | cut -c1-300
