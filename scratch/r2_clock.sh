#!/bin/bash
# scratch/r2_clock.sh <variant tags...> -- shader clock under load: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs)
# against the dispatch duration of the dominant scale kernel, per library variant
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r2clock; mkdir -p "$out"
for v in "$@"; do
  lib=$PWD/timg_amd/libtimg_hip_$v.so; [ "$v" = base ] && lib=$PWD/timg_amd/libtimg_hip.so
  d="$out/$v"; rm -rf "$d"; mkdir -p "$d"
  TIMG_HIP_LIB=$lib ROUNDS=3 timeout 180 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$d" -o pmc -- python scratch/bench_scale.py > "$d/log.txt" 2>&1 || tail -3 "$d/log.txt"
  python3 - "$d" "$v" <<'PY' | tee -a "$out/clock.txt"
import csv, sys, glob, statistics
d, v = sys.argv[1], sys.argv[2]
cyc = {}
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "ScaleStreamMKernel<0" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc[r["Dispatch_Id"]] = float(r["Counter_Value"])
dur = {}
for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "ScaleStreamMKernel<0" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
ids = sorted(set(cyc) & set(dur), key=int)[-40:]
if ids:
    c = statistics.median(cyc[i] for i in ids) / 8.0
    t = statistics.median(dur[i] for i in ids)
    print(f"{v}: {t/1e3:.1f} us per launch, {c:.0f} cycles per XCD -> {c/t:.3f} GHz")
else:
    print(v, "no dispatches found", len(cyc), len(dur))
PY
  find "$d" -name '*.csv' -delete
done
