#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
for v in ${VARS:-base pb2 pb8 pb16 base}; do
  lib=$PWD/timg_amd/libtimg_hip_$v.so; [ $v = base ] && lib=$PWD/timg_amd/libtimg_hip.so
  d=gpurun_out/cw_$v; rm -rf $d; mkdir -p $d
  TIMG_HIP_LIB=$lib HS=450 REPS=12 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python scratch/dither_rows.py > $d/log.txt 2>&1
  python3 - $d $v <<'PY'
import csv, glob, sys
fn = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(fn)):
    if "MedianCut" in r["Name"] or "DitherKernel" in r["Name"]:
        import re
        print(sys.argv[2], re.search(r"(\w+Kernel)", r["Name"]).group(1), r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1))
PY
  find $d -name '*.csv' -delete
done
