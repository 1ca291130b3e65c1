#!/bin/bash
# scratch/r2_pmc2.sh <variant tags...> -- SQ busy/wait split of ScaleStreamMKernel<0> per library variant
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r2pmc2; mkdir -p "$out"
for v in "$@"; do
  lib=$PWD/timg_amd/libtimg_hip_$v.so; [ "$v" = base ] && lib=$PWD/timg_amd/libtimg_hip.so
  d="$out/$v"; rm -rf "$d"; mkdir -p "$d"
  TIMG_HIP_LIB=$lib ROUNDS=2 REPS=5 WARM_S=0.1 timeout 180 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d "$d" -o pmc -- python scratch/bench_scale.py > "$d/log.txt" 2>&1 || tail -3 "$d/log.txt"
  python3 - "$d" "$v" <<'PY' | tee -a "$out/pmc.txt"
import csv, sys, glob, collections
d, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "ScaleStreamMKernel<0" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(v, {c: round(sum(x) / len(x) / 1e6, 1) for c, x in acc.items()}, "(millions)")
PY
  find "$d" -name '*.csv' -delete
done
