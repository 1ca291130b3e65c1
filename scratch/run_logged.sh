#!/bin/bash
# scratch/run_logged.sh <name> <command...> -- runs a command on the GPU box with its whole output kept under
# gpurun_out/${TIMG_ROUND:-r4}/<name>.log; when it dies of a signal (rc >= 128: a GPU memory fault aborts the process) the box
# canary (pure torch, none of this repository's code) runs right after it on the same lease, so that the log says
# whether the BOX or the code was at fault.  (One lease in round 3 crashed three different processes of this
# repository in a row; the same binaries were clean on the leases before and after it, and nothing was kept.)
name=$1; shift
mkdir -p gpurun_out/${TIMG_ROUND:-r4}
"$@" > gpurun_out/${TIMG_ROUND:-r4}/$name.log 2>&1
rc=$?
echo "[$name] rc=$rc"
if [ $rc -ge 128 ]; then
  echo "[$name] died of a signal: last lines and the canary's verdict follow"
  grep -v "^Extension modules" gpurun_out/${TIMG_ROUND:-r4}/$name.log | tail -40 | cut -c1-300
  timeout 600 python3 tests/box_canary.py > gpurun_out/${TIMG_ROUND:-r4}/$name.canary.log 2>&1
  echo "[$name] canary rc=$?"; tail -5 gpurun_out/${TIMG_ROUND:-r4}/$name.canary.log | cut -c1-300
fi
exit $rc
