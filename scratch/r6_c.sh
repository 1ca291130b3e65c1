#!/bin/bash
# scratch/r6_c.sh -- c4 / c5 at other frames-per-launch: does a longer batch shorten the latency-bound sixel chain?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
out=gpurun_out/r6; mkdir -p "$out"
: > "$out/chunk_sweep.txt"
for c in 64 100 150 200 300; do
  echo "== c4 --chunk $c" >> "$out/chunk_sweep.txt"
  timeout 300 python bench.py --config c4 --chunk $c --steps 5 --warmup 2 --no-extras 2>>"$out/chunk_sweep.err" | tail -1 >> "$out/chunk_sweep.txt"
done
for c in 64 128 256; do
  echo "== c5 --chunk $c" >> "$out/chunk_sweep.txt"
  timeout 400 python bench.py --config c5 --chunk $c --steps 3 --warmup 1 --no-extras 2>>"$out/chunk_sweep.err" | tail -1 >> "$out/chunk_sweep.txt"
done
python - <<'PY'
import json
for line in open("gpurun_out/r6/chunk_sweep.txt"):
    line = line.strip()
    if line.startswith("=="):
        print(line, end="  ")
        continue
    try:
        d = json.loads(line)
        print("ms_per_step", d["ms_per_step"], "value", d["value"], "encode", d.get("encode_ms_per_step"), "parity", d.get("parity_check"))
    except Exception as e:
        print("??", line[:200])
PY
tail -5 "$out/chunk_sweep.err"
