#!/bin/bash
# scratch/r2_pmc.sh <outname> [env...] -- SQ counter passes over scratch/bench_scale.py (all ScaleStream kernels)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
name=$1; shift
out=gpurun_out/$name; rm -rf "$out"; mkdir -p "$out"
env "$@" N=64 timeout 120 python scratch/bench_scale.py 2>&1 | grep "^kernel" | tee "$out/time.txt"
pmc() {
  d="$out/$1"; mkdir -p "$d"; shift
  env "$@" N=64 timeout 180 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$d" -o pmc -- python scratch/bench_scale.py > "$d/log.txt" 2>&1 || tail -5 "$d/log.txt"
  f=$(find "$d" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a "$out/pmc.txt"
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "ScaleStream" not in n: continue
    k = "ScaleStream" + n.split("ScaleStream")[1][:12]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    if max(d.get("SQ_WAVES", [1e9])) < 1 : continue
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
  find "$d" -name '*.csv' -delete
}
CTRS="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" pmc sq1 "$@"
CTRS="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" pmc sq2 "$@"
CTRS="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_TRANS" pmc sq3 "$@"
