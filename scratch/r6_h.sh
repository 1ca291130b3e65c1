#!/bin/bash
# scratch/r6_h.sh -- (1) the drop-in path after the loader slots became a free-list (c2 must be back under a millisecond a frame);
# (2) SQ counters of the SIXEL kernels of the default bench step (never collected before: profiles/*/sq_counters_* hold the scale
# kernels only), one counter set per run, --kernel-trace only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
E=tests/twins/build/twin_bench
$E --config c2,metric --repeat 3 --paths gpu 2>/dev/null | grep '^{' > "$out/twin_bench_slots.txt"
python3 - <<'PY'
import json
for l in open("gpurun_out/r6/twin_bench_slots.txt"):
    d = json.loads(l); print(d["config"], d["path"], "queue", d["queue_len"], "%.1f Gpx/s" % (d["mpx_per_s"] / 1e3), "%.3f ms/frame" % d["ms_per_frame"])
PY
args="--no-cpu-baseline --no-extras --no-dropin --no-parity --steps 3 --warmup 1"
w=$out/pmc_sixel; rm -rf "$w"; mkdir -p "$w"
pass() { d="$w/$1"; shift
  timeout -k 5 170 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$d" -o pmc -- python bench.py $args > "$d.log" 2>&1 || { echo "pass $d failed"; tail -3 "$d.log"; }
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass sq3 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_LEVEL_LDS
python3 - "$w" "$out" <<'PY'
import csv, glob, sys, collections
w, out = sys.argv[1:3]
import re
def short(n):
    m = re.search(r"(\w+Kernel(?:<[^>]*>)?)", n)
    return m.group(1) if m else n.split("(")[0]
def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(w + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
def durations(sub):
    acc = collections.defaultdict(list)
    for fn in glob.glob(w + "/" + sub + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            acc[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
s1, s2, s3, d1 = counters("sq1"), counters("sq2"), counters("sq3"), durations("sq1")
with open(out + "/sq_counters_sixel.txt", "w") as f:
    f.write("# python bench.py (default step: 64 frames 800x450 -> sixel) under rocprofv3 --kernel-trace --pmc, three counter sets in three runs\n"
            "# (scratch/r6_h.sh): averages per dispatch; us = the kernel's duration in the first counter run\n")
    for k in sorted(s1, key=lambda k: -d1.get(k, 0)):
        if "ScaleStream" in k: continue
        f.write("%s  %.1f us\n  sq1 %s\n  sq2 %s\n  sq3 %s\n" % (k, d1.get(k, 0) / 1e3, {c: round(v, 1) for c, v in s1[k].items()},
                {c: round(v, 1) for c, v in s2.get(k, {}).items()}, {c: round(v, 1) for c, v in s3.get(k, {}).items()}))
print(open(out + "/sq_counters_sixel.txt").read()[:6000])
PY
rm -rf "$w"
