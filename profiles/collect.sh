#!/bin/bash
# profiles/collect.sh <round-tag>   (run on the GPU box: gpurun -- 'bash profiles/collect.sh r1')
#
# 1. rocprofv3 --kernel-trace --stats of the default bench.py command
#      -> gpurun_out/<tag>/kernel_stats.csv, summary.txt   (bench.py --no-extras: the timed region only)
# 2. rocprofv3 --pmc FETCH_SIZE and (separate pass) --pmc WRITE_SIZE of the same command
#      -> gpurun_out/<tag>/hbm_traffic.json   (copy to profiles/hbm_traffic.json: bench.py reads it)
# Counters are collected in their own runs (never together with trace domains other than
# --kernel-trace).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts the
# 128-byte requests of wide coalesced reads (16 B/lane, what the scale kernel issues) as 64 bytes,
# so it is doubled; unit of both counters: KiB.
set -e
tag=${1:-r1}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; rm -rf "$out"; mkdir -p "$out"
ulimit -c 0   # (a crashing run must not spend minutes dumping a core of the GPU mappings)
args="--no-cpu-baseline --no-extras --steps 5 --warmup 2 $*"

timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o prof -- python bench.py $args > "$out/bench_trace.log" 2>&1
cp "$(find "$out/trace" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o pmc -- python bench.py $args > "$out/bench_fetch.log" 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -o pmc -- python bench.py $args > "$out/bench_write.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/kernel_stats.csv")))
with open(out + "/summary.txt", "w") as f:
    f.write("%-34s %6s %12s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in rows:
        n = r["Name"]
        if "timg_amd" not in n: continue
        short = n.split("(anonymous namespace)::")[1].split("(")[0] if "(anonymous namespace)::" in n else n.split("(")[0]
        if "ScaleStream" in n: short = "ScaleStream" + n.split("ScaleStream")[1].split("(")[0]
        f.write("%-34s %6s %12.1f %12.1f %7.2f\n" % (short[:34], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                  float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
print(open(out + "/summary.txt").read())
def counter(sub, name):
    acc = collections.defaultdict(list)
    for fn in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == name and "ScaleStreamMKernel" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("ScaleStreamMKernel")[1][:2] + ">"].append(float(r["Counter_Value"]))  # "<0>" / "<1>"
    return {k: sum(v) / len(v) for k, v in acc.items()}
fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
k = "<0>"
res = {
    "kernel": "streaming", "config": "metric", "workload_frames": 64,
    "dominant_kernel": "ScaleStreamMKernel<0> (opaque channel set, vertical products on the matrix cores)",
    "FETCH_SIZE_KiB_raw": fetch.get(k), "WRITE_SIZE_KiB_raw": write.get(k),
    "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests of 16-B/lane reads as 64 B), WRITE_SIZE as reported; KiB",
    "hbm_bytes_per_launch": int((fetch.get(k, 0) * 2 + write.get(k, 0)) * 1024),
}
json.dump(res, open(out + "/hbm_traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
tail -1 "$out/bench_trace.log" | cut -c1-300
rm -rf "$out/trace" "$out/fetch" "$out/write"
