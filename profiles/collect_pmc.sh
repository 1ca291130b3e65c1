#!/bin/bash
# profiles/collect_pmc.sh <round-tag> [config[:kind] ...]   (run on the GPU box: gpurun -- 'bash profiles/collect_pmc.sh r5')
#
# Counter evidence for the scale kernel of every bench configuration, as the kernels are NOW: for each configuration
# (default: metric, metric:alpha, c3, c5) the same `python bench.py --config C [--kind K]` command under
#   rocprofv3 --kernel-trace --pmc FETCH_SIZE            } HBM traffic of the dominant ScaleStream kernel, corrected as
#   rocprofv3 --kernel-trace --pmc WRITE_SIZE            } MI355X_MICROARCH.md prescribes (profiles/collect.sh)
#   rocprofv3 --kernel-trace --pmc <SQ set 1> / <SQ set 2>  busy / wait split, instruction counts by kind, LDS conflicts
# every counter set in its own run, never with trace domains other than --kernel-trace.  Results:
#   gpurun_out/<tag>/hbm_traffic_<config>[_<kind>].json   (copy to profiles/: bench.py fills roofline.traffic and
#                                                          roofline.limiter of `--config C [--kind K]` from it)
#   gpurun_out/<tag>/sq_counters_<config>[_<kind>].txt
# Every hbm_traffic_*.json carries the SHA-256 of the timg_amd/libtimg_hip.so it was collected with: bench.py compares it with
# the library it is timing and says `traffic_stale: true` (and drops `limiter`) when a kernel change has left the file behind.
set -e
tag=${1:-r4}; shift || true
specs=("$@"); [ ${#specs[@]} -eq 0 ] && specs=(metric metric:alpha c3 c5)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p "$out"
ulimit -c 0
for spec in "${specs[@]}"; do
  cfg=${spec%%:*}; kind=; [ "$spec" != "$cfg" ] && kind=${spec#*:}
  name=$cfg; [ -n "$kind" ] && name=${cfg}_$kind
  args="--config $cfg --no-cpu-baseline --no-extras --no-dropin --no-parity --steps 3 --warmup 1"
  [ -n "$kind" ] && args="$args --kind $kind"
  w=$out/pmc_$name; rm -rf "$w"; mkdir -p "$w"
  timeout -k 5 150 python bench.py $args > "$w/bench.json" 2> "$w/bench.err" || { echo "$name: bench failed"; tail -3 "$w/bench.err"; continue; }
  pass() {  # <subdir> <counters...>
    d="$w/$1"; shift
    timeout -k 5 170 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$d" -o pmc -- python bench.py $args > "$d.log" 2>&1 || { echo "$name: pass $d failed"; tail -3 "$d.log"; }
  }
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
  pass sq3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VALU_CVT
  pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
  python3 - "$w" "$out" "$name" "$cfg" "$kind" <<'PY'
import csv, glob, json, sys, collections
w, out, name, cfg, kind = sys.argv[1:6]
line = json.loads([l for l in open(w + "/bench.json") if l.startswith("{")][-1])
def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(w + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "ScaleStream" in r["Kernel_Name"]:
                k = "ScaleStream" + r["Kernel_Name"].split("ScaleStream")[1].split("(")[0]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
sq1, sq2, sq3, fetch, write = counters("sq1"), counters("sq2"), counters("sq3"), counters("fetch"), counters("write")
def durations(sub):
    """average duration (ns) per kernel of the pass's own kernel trace"""
    acc = collections.defaultdict(list)
    for fn in glob.glob(w + "/" + sub + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "ScaleStream" in r["Kernel_Name"]:
                k = "ScaleStream" + r["Kernel_Name"].split("ScaleStream")[1].split("(")[0]
                acc[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
dur3 = durations("sq3")
# the dominant kernel: most wave cycles per dispatch
dom = max(sq1, key=lambda k: sq1[k].get("SQ_WAVE_CYCLES", 0.0)) if sq1 else (max(fetch, key=lambda k: fetch[k].get("FETCH_SIZE", 0.0)) if fetch else None)
with open("%s/sq_counters_%s.txt" % (out, name), "w") as f:
    f.write("# python bench.py --config %s%s under rocprofv3 --pmc (profiles/collect_pmc.sh): averages per dispatch\n" % (cfg, " --kind " + kind if kind else ""))
    for k in sorted(set(sq1) | set(sq2)):
        f.write("%s%s\n  sq1 %s\n  sq2 %s\n  sq3 %s\n" % (k, "   <- dominant" if k == dom else "", {c: round(v, 1) for c, v in sq1.get(k, {}).items()},
                                              {c: round(v, 1) for c, v in sq2.get(k, {}).items()}, {c: round(v, 1) for c, v in sq3.get(k, {}).items()}))
import hashlib, os
lib = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "timg_amd", "libtimg_hip.so")
res = {"library_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),  # (bench.py: a counter file of another binary is stale)
       "config": cfg, "kind": kind or None, "workload_frames": line["config"]["frames_per_launch"], "kernel": line["config"]["scale_kernel"],
       "dominant_kernel": dom, "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"],
       "avg_launch_ms_unprofiled": line["roofline"]["avg_launch_ms"], "frac_unprofiled": line["roofline"]["frac"]}
if dom and dom in fetch and dom in write:
    fr, wr = fetch[dom]["FETCH_SIZE"], write[dom]["WRITE_SIZE"]
    res.update({"FETCH_SIZE_KiB_raw": fr, "WRITE_SIZE_KiB_raw": wr,
                "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests of 16-B/lane reads as 64 B), WRITE_SIZE as reported; KiB",
                "hbm_bytes_per_launch": int((fr * 2 + wr) * 1024)})
    res["traffic_over_algorithmic"] = round(res["hbm_bytes_per_launch"] / res["algorithmic_bytes_per_launch"], 3)
if dom and dom in sq1:
    a, b = sq1[dom], sq2.get(dom, {})
    wc, waves = a.get("SQ_WAVE_CYCLES", 0.0), max(1.0, a.get("SQ_WAVES", 1.0))
    if wc > 0:
        res["sq"] = {"waves": waves, "wave_cycles_per_wave": round(wc / waves, 1),
                     "issuing_frac": round(a.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3), "waiting_frac": round(a.get("SQ_WAIT_ANY", 0.0) / wc, 3),
                     "issue_stalled_frac": round(a.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
                     "valu_per_wave": round(a.get("SQ_INSTS_VALU", 0.0) / waves, 1), "salu_per_wave": round(b.get("SQ_INSTS_SALU", 0.0) / waves, 1),
                     "lds_per_wave": round(b.get("SQ_INSTS_LDS", 0.0) / waves, 1), "vmem_rd_per_wave": round(b.get("SQ_INSTS_VMEM_RD", 0.0) / waves, 1),
                     "lds_bank_conflict_over_active": round(b.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, b.get("SQ_LDS_IDX_ACTIVE", 1.0)), 3)}
        # The OTHER roof of a byte-shuffling kernel: what its vector instructions cost to ISSUE.  A wave-wide VALU (or
        # v_mfma_f32_4x4x1) instruction holds its SIMD for 4 clocks; the chip has 4 SIMDs in each of 256 CUs; the clock is
        # the one the counters saw (GRBM_GUI_ACTIVE summed over the 8 XCDs / the dispatch's duration in that pass).
        c = sq3.get(dom, {})
        if c.get("GRBM_GUI_ACTIVE") and dur3.get(dom):
            clock_ghz = c["GRBM_GUI_ACTIVE"] / 8.0 / dur3[dom]
            valu = a.get("SQ_INSTS_VALU", 0.0)
            issue_ms = valu * 4.0 / 1024.0 / (clock_ghz * 1e6)
            res["valu_issue"] = {"vector_instructions_per_wave": round(valu / waves, 1), "mfma_per_wave": round(c.get("SQ_INSTS_MFMA", 0.0) / waves, 1),
                                 "cvt_per_wave": round(c.get("SQ_INSTS_VALU_CVT", 0.0) / waves, 1),
                                 "clock_ghz_under_counters": round(clock_ghz, 3), "issue_ms_per_launch": round(issue_ms, 4),
                                 "launch_ms_in_that_pass": round(dur3[dom] * 1e-6, 4), "frac_of_launch": round(issue_ms / (dur3[dom] * 1e-6), 3),
                                 "how": "SQ_INSTS_VALU x 4 clocks / (256 CUs x 4 SIMDs) / clock; clock = GRBM_GUI_ACTIVE / 8 XCDs / duration"}
        s = res["sq"]
        res["limiter"] = ("profiles/%s/sq_counters_%s.txt (%s): waves issue %.0f %% of their cycles, wait (s_waitcnt / barrier) %.0f %%, "
                          "are issue-stalled %.0f %%; %d VALU + %d SALU + %d LDS + %d VMEM-read instructions per wave; HBM traffic %s x algorithmic"
                          % (out.split("/")[-1], name, dom, 100 * s["issuing_frac"], 100 * s["waiting_frac"], 100 * s["issue_stalled_frac"],
                             s["valu_per_wave"], s["salu_per_wave"], s["lds_per_wave"], s["vmem_rd_per_wave"], res.get("traffic_over_algorithmic", "?")))
json.dump(res, open("%s/hbm_traffic_%s.json" % (out, name), "w"), indent=1)
print(name, json.dumps({k: res[k] for k in res if k in ("dominant_kernel", "hbm_bytes_per_launch", "traffic_over_algorithmic", "frac_unprofiled", "limiter")})[:600])
PY
  rm -rf "$w"/fetch "$w"/write "$w"/sq1 "$w"/sq2 "$w"/sq3
done
