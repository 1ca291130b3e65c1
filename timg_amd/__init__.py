"""timg_amd -- MI355X-native twin of hzeller/timg's rendering hot path.

The product is ``libtimg_hip.so`` (hand-written HIP for gfx950 behind the C-ABI
of ``include/timg_hip.h``) plus the C++ twins in ``timg_amd/twins``.  This
Python package is plumbing only: a ctypes binding used by the tests and by
``bench.py``.  It never falls back to a CPU implementation -- if the shared
library is missing or no GPU is usable, calls raise.
"""
from .hip import TimgHip, TimgHipError, Blend, lib_path, load_library  # noqa: F401
