#!/bin/bash
# scratch/r5_r.sh -- is a sequential read of librccl.so.1 what a cold box needs before the gather stage?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
mkdir -p gpurun_out/r5
{
f=$(readlink -f /opt/rocm/lib/librccl.so.1)
grep -E "^(Cached|MemFree)" /proc/meminfo
python3 - "$f" <<'PY'
import os, sys, mmap, time
f = sys.argv[1]
fd = os.open(f, os.O_RDONLY)
n = os.fstat(fd).st_size
m = mmap.mmap(fd, n, prot=mmap.PROT_READ)
try:
    import ctypes
    libc = ctypes.CDLL(None, use_errno=True)
    vec = (ctypes.c_ubyte * ((n + 4095) // 4096))()
    addr = ctypes.addressof(ctypes.c_char.from_buffer_copy(b"x"))  # placeholder
except Exception as e:
    print("mincore setup:", e)
t = time.time()
tot = 0
with open(f, "rb") as h:
    while True:
        b = h.read(16 << 20)
        if not b: break
        tot += len(b)
print("sequential read of %s: %d MB in %.1f s" % (f, tot >> 20, time.time() - t))
PY
s=$(date +%s)
timeout 700 tests/twins/build/twin_check gather 2>&1 | tail -2 | cut -c1-160
echo "twin_check gather: $(( $(date +%s) - s )) s"
s=$(date +%s)
timeout 900 tests/twins/build/twin_check all /tmp/dump.bin 2>&1 | tail -1 | cut -c1-160
echo "twin_check all: $(( $(date +%s) - s )) s"
} > gpurun_out/r5/twin_times.txt 2>&1
cat gpurun_out/r5/twin_times.txt
