#!/bin/bash
# scratch/r4_dither_sweep.sh [lib-tag[:VAR=VALUE] ...] -- median per-dispatch time of the sixel kernels for 64 frames of each
# geometry in $GEOMS, for the main library and each variant
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
timeout 120 python3 tests/box_canary.py > gpurun_out/r4_canary.log 2>&1; echo "canary rc=$? $(tail -1 gpurun_out/r4_canary.log | cut -c1-200)"
export GEOMS=${GEOMS:-800x96,400x96,800x450,400x450}
for spec in main "$@"; do
  tag=${spec%%:*}; extra=; [ "$spec" != "$tag" ] && extra=${spec#*:}
  name=$(echo "$spec" | tr ':=' '__')
  d=gpurun_out/r4/sweep_$name; rm -rf $d; mkdir -p $d
  lib=; [ $tag != main ] && lib="TIMG_HIP_LIB=$GRAFT_REPO_ROOT/timg_amd/libtimg_hip_$tag.so"
  env $lib $extra timeout -k 5 90 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python scratch/r4_dither_sweep.py > $d/log.txt 2>&1 || tail -5 $d/log.txt
  echo "== $spec"
  python3 - $d <<'PY' | tee $d/result.txt
import csv, glob, sys, collections, re, os
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        m = re.search(r"(\w+Kernel)", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:30]))
rows.sort()
seq = collections.defaultdict(list)
for s, e, k in rows:
    seq[k].append((e - s) / 1000.0)
geoms = os.environ["GEOMS"].split(",")
for k, v in seq.items():
    if len(v) == 5 * len(geoms):
        print("%-20s" % k, "  ".join("%s: %.1f" % (g, sorted(v[i * 5:(i + 1) * 5])[2]) for i, g in enumerate(geoms)))
PY
  find $d -name '*.csv' -delete
done
