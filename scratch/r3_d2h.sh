#!/bin/bash
for q in 8 4; do
GPU_MAX_HW_QUEUES=$q timeout 600 python3 bench.py --no-cpu-baseline --no-dropin --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('queues $q: ms/step',d['ms_per_step'],'d2h',d['ms_per_step_with_d2h'],'overlapped',d['ms_per_step_with_d2h_overlapped'],'batched',d['batched_streams']['ms_per_step'])"
done
python3 - <<'PY'
import torch, ctypes
t=torch.empty(1<<20,dtype=torch.uint8).pin_memory()
hip=ctypes.CDLL('libamdhip64.so')
class Attr(ctypes.Structure):
    _fields_=[('type',ctypes.c_int),('device',ctypes.c_int),('devicePointer',ctypes.c_void_p),('hostPointer',ctypes.c_void_p),('isManaged',ctypes.c_int),('allocationFlags',ctypes.c_uint)]
a=Attr()
rc=hip.hipPointerGetAttributes(ctypes.byref(a), ctypes.c_void_p(t.data_ptr()))
print('pinned torch tensor: rc',rc,'type',a.type)
PY
