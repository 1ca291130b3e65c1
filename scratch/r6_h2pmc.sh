#!/bin/bash
# scratch/r6_h2pmc.sh [lib-tag] -- SQ counters of the horizontal-first scale kernels (two columns a lane pair vs one), 16 8K S-alpha frames
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
lib=$PWD/timg_amd/libtimg_hip.so; [ -n "$1" ] && lib=$PWD/timg_amd/libtimg_hip_$1.so
for h2 in 1 0; do
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
    d=/tmp/h2pmc_$h2; rm -rf $d
    TIMG_HIP_LIB=$lib TIMG_HIP_H2=$h2 N=16 SW=7680 SH=4320 KIND=alpha ROUNDS=2 REPS=3 WARM_S=0.1 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o pmc -- python scratch/bench_scale.py > /tmp/h2pmc.log 2>&1
    python3 - $d $h2 <<'PY'
import csv, glob, sys, collections
d, h2 = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "ScaleStreamH" in r["Kernel_Name"]:
            k = "ScaleStreamH" + r["Kernel_Name"].split("ScaleStreamH")[1].split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "ScaleStreamH" in r["Kernel_Name"]:
            k = "ScaleStreamH" + r["Kernel_Name"].split("ScaleStreamH")[1].split("(")[0]
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in sorted(acc):
    print("H2=%s %s  %.3f ms  %s" % (h2, k, sum(dur[k]) / len(dur[k]) * 1e-6, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in acc[k].items()}), "(counters in millions)")
PY
  done
done > "$out/h2_pmc.txt" 2>&1
cat "$out/h2_pmc.txt"
