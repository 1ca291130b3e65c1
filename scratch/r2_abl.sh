#!/bin/bash
# scratch/r2_abl.sh -- ablation timings of the old streaming kernel + extended VALU microbenchmark
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r2abl; rm -rf "$out"; mkdir -p "$out"
timeout 60 scratch/ubench/valu_rate.bin > "$out/valu_rate.txt" 2>&1; cat "$out/valu_rate.txt"
for v in "" _nobar _nohoriz; do
  lib=$PWD/timg_amd/libtimg_hip$v.so
  echo "== lib$v" | tee -a "$out/abl.txt"
  TIMG_HIP_LIB=$lib N=64 KIND=photo timeout 120 python scratch/bench_scale.py 2>&1 | grep "^kernel" | tee -a "$out/abl.txt"
done
d="$out/lvl"; mkdir -p "$d"
N=64 timeout 180 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES --output-format csv -d "$d" -o pmc -- python scratch/bench_scale.py > "$d/log.txt" 2>&1 || tail -5 "$d/log.txt"
f=$(find "$d" -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY' | tee -a "$out/pmc.txt"
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "ScaleStreamKernel<0>" not in n: continue
    acc["k0"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, dd in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in dd.items()})
PY
find "$d" -name '*.csv' -delete
