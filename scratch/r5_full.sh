#!/bin/bash
# scratch/r5_full.sh -- the driver's two commands: pytest -m gpu (whole suite) and smoke()
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
mkdir -p gpurun_out/r5
TIMG_ROUND=r5 scratch/run_logged.sh pytest timeout -k 10 1500 python3 -X faulthandler -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=12
tail -25 gpurun_out/r5/pytest.log | cut -c1-300
TIMG_ROUND=r5 scratch/run_logged.sh smoke timeout -k 5 300 python3 -c "import __graft_entry__ as g; g.smoke()"
tail -3 gpurun_out/r5/smoke.log | cut -c1-300
