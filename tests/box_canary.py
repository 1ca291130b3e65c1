"""GPU-box canary: a child process that touches the GPU with NOTHING of this repository -- pure
torch: device allocation, a kernel, pageable and pinned host<->device copies -- so that a box whose GPU
faults on first use (round 2's driver lease: RCCL's own init aborted and a 2.5-second smoke run died
with `Memory access fault by GPU` before/while the first copy ran) is told apart from a fault of this
repository's kernels.  Used by tests/conftest.py (session start of `-m gpu` runs) and by
__graft_entry__.smoke().  Retries: a fresh lease may still be settling after a previous tenant's reset."""
import subprocess
import sys
import time

_CODE = r'''
import torch, sys
assert torch.cuda.is_available(), "no GPU"
d = torch.device("cuda:0")
x = torch.arange(1 << 22, device=d, dtype=torch.int32)
h = (torch.arange(1 << 20, dtype=torch.int32) * 3)             # pageable host memory
p = torch.empty(1 << 20, dtype=torch.int32).pin_memory()       # pinned host memory
y = h.to(d) + x[: 1 << 20]
p.copy_(y, non_blocking=True); torch.cuda.synchronize()
assert int(p[12345]) == 12345 * 4 and int(y.sum().item()) == 4 * ((1 << 20) - 1) * (1 << 20) // 2
props = torch.cuda.get_device_properties(0)
print("canary ok:", props.name, props.gcnArchName, props.multi_processor_count, "CUs,", props.total_memory >> 30, "GiB,",
      "hip", torch.version.hip)
'''


def run_canary(attempts: int = 3, pause_s: float = 15.0, timeout_s: float = 420.0):
    """Returns (ok, text).  text = the canary's line, or every failed attempt's tail."""
    notes = []
    for k in range(attempts):
        try:
            r = subprocess.run([sys.executable, "-c", _CODE], capture_output=True, text=True, timeout=timeout_s)
            out = (r.stdout + r.stderr).strip()
            if r.returncode == 0:
                line = [ln for ln in out.splitlines() if ln.startswith("canary ok")][-1:]
                if k:
                    notes.append(f"attempt {k + 1}: ok")
                return True, "; ".join(notes + line)
            notes.append(f"attempt {k + 1}: rc {r.returncode}: {out[-400:]}")
        except subprocess.TimeoutExpired:
            notes.append(f"attempt {k + 1}: timed out after {timeout_s:.0f} s")
        if k + 1 < attempts:
            time.sleep(pause_s)
    return False, " | ".join(notes)


if __name__ == "__main__":
    ok, text = run_canary()
    print(("GPU BOX OK: " if ok else "GPU BOX UNHEALTHY (pure torch, none of this repository's code): ") + text)
    sys.exit(0 if ok else 1)
