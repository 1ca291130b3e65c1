#!/bin/bash
# scratch/r2_evidence.sh -- round-2 evidence for the scale kernel (run through gpurun):
# VALU issue-rate microbenchmark, band-height sweep, SQ counters of ScaleStreamKernel<0>.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r2ev; rm -rf "$out"; mkdir -p "$out"
( cd scratch/ubench && hipcc --offload-arch=gfx950 -O3 -o valu_rate.bin valu_rate.hip 2>&1 | tail -3; timeout 60 ./valu_rate.bin ) > "$out/valu_rate.txt" 2>&1
cat "$out/valu_rate.txt"
for b in 45 50 75 150; do
  TIMG_HIP_BAND_ROWS=$b N=64 KIND=photo timeout 120 python scratch/bench_scale.py 2>&1 | grep "^kernel" | tee -a "$out/band_sweep.txt"
done
for b in 45 150; do
  TIMG_HIP_BAND_ROWS=$b N=64 KIND=alpha timeout 120 python scratch/bench_scale.py 2>&1 | grep "^kernel" | tee -a "$out/band_sweep.txt"
done
pmc() {  # name, counters
  d="$out/$1"; mkdir -p "$d"
  N=64 KIND=${KIND:-photo} timeout 180 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d "$d" -o pmc -- python scratch/bench_scale.py > "$d/log.txt" 2>&1 || tail -5 "$d/log.txt"
  f=$(find "$d" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" "$1" <<'PY' | tee -a "$out/pmc.txt"
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "ScaleStream" not in n: continue
    k = "ScaleStream" + n.split("ScaleStream")[1][:12]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(sys.argv[2], k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
  find "$d" -name '*.csv' -delete
}
pmc sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
pmc sq2 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
pmc sq3 "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_CYCLES"
KIND=alpha pmc sq1_alpha "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > "$out/sq_counter_names.txt"
