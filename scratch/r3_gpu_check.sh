#!/bin/bash
# the driver's two round-end commands, verbatim, in fresh processes; logs under gpurun_out/r3/
O=gpurun_out/r3; mkdir -p $O
rocminfo 2>/dev/null | grep -E "^\*\*\*|Agent [0-9]|Marketing Name|Node:|Compute Unit|Xnack|Partition" > $O/rocminfo.txt
env | grep -E "HSA|HIP|ROCR|NCCL|RCCL|AMD_|GPU_" > $O/env.txt
timeout 1700 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
timeout 600 python3 -c 'import sys; sys.path.insert(0, "."); import __graft_entry__ as e
f = getattr(e, "smoke", None)
f(); print("__SMOKE_OK__")' > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
tail -n 15 $O/pytest.log; tail -n 5 $O/smoke.log
