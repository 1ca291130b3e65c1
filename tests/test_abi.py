"""CPU-only: the C-ABI library loads without a GPU, exports every symbol
include/timg_hip.h declares, sizes its buffers like the reference, and fails
loudly (no CPU fallback) when no device is present."""
import ctypes
import os
import re

import pytest

import timg_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "timg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(timg_hip_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    lib = timg_amd.load_library()
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/timg_hip.h but not exported"


def test_comm_library_exports_its_header_and_shards_like_the_python_path():
    """include/timg_hip_comm.h (the RCCL gather, its own library): every declared symbol is exported;
    the frame -> (rank, index) arithmetic the C++ writer uses is the inverse of shard_frames."""
    from timg_amd import comm
    from timg_amd.gather import shard_frames
    text = open(os.path.join(ROOT, "include", "timg_hip_comm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(timg_hip_[a-z0-9_]+)\s*\(", text)))
    lib = comm.load_comm_library()
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), n
    for n_total, world in [(600, 8), (256, 8), (12, 4), (7, 3), (5, 8), (1, 1), (64, 2)]:
        for rr in (False, True):
            owned = [shard_frames(n_total, world, r, round_robin=rr) for r in range(world)]
            for r in range(world):
                assert comm.shard_count(n_total, world, rr, r) == len(owned[r])
            for f in range(n_total):
                r, i = comm.shard_locate(n_total, world, rr, f)
                assert owned[r][i] == f, (n_total, world, rr, f)
    # no communicator without a usable device / peer: errors are reported, nothing crashes
    assert lib.timg_hip_comm_create(0, 0, 0, None, None) == -1
    lib.timg_hip_comm_destroy(None)


def test_version_and_buffer_size_rules(oracle):
    lib = timg_amd.load_library()
    assert lib.timg_hip_version() >> 16 == 1
    for w, h in [(1, 1), (67, 50), (100, 56), (801, 451)]:
        assert lib.timg_hip_block_max_bytes(w, h) == oracle.block_max_bytes(w, h)
        r6 = (h + 5) - (h + 5) % 6  # round_to_sixel, src/sixel-canvas.cc:91-94
        assert lib.timg_hip_sixel_max_bytes(w, h) == 1024 + w * r6 * 5


def test_init_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(timg_amd.TimgHipError) as e:
        timg_amd.TimgHip(0)
    assert e.value.code == -2 and "device" in str(e.value).lower()


def test_null_arguments_are_rejected_not_crashing():
    lib = timg_amd.load_library()
    assert lib.timg_hip_init(0, None) == -1
    assert lib.timg_hip_scaler_create(None, 1, 1, 0, 1, 1, 0, None) == -1
    assert lib.timg_hip_sync(None, None) == -1
    lib.timg_hip_destroy(None)
    lib.timg_hip_scaler_destroy(None)


def test_product_does_not_reference_the_oracle():
    """The shipped library must not link or dlopen anything under oracle/."""
    needed = os.popen(f"readelf -d {timg_amd.lib_path()} 2>/dev/null").read()
    assert "oracle" not in needed and "timg_ref" not in needed
    for dirpath, _, files in os.walk(os.path.join(ROOT, "timg_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cc", ".h")):
                body = open(os.path.join(dirpath, f), errors="replace").read()
                assert "libtimg_oracle" not in body and "oracle_lib" not in body, f


def test_product_library_has_no_debug_entry_points():
    """Host-side emulations used by the CPU tests live in libtimg_hip_debug.so, not in the product."""
    syms = os.popen(f"nm -D --defined-only {timg_amd.lib_path()} 2>/dev/null").read()
    assert "timg_hip_init" in syms and "timg_hip_debug" not in syms


def test_header_is_valid_c99():
    """The drop-in boundary is a C header: it must compile as C, not only as C++."""
    import subprocess
    src = '#include "timg_hip.h"\nint main(void) { return timg_hip_version() ? 0 : 1; }\n'
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I",
                        os.path.join(ROOT, "include"), "-x", "c", "-"], input=src, text=True, capture_output=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_c_example_runs_and_matches_the_python_binding(oracle):
    """examples/abi_demo.c (plain C against the C-ABI): its sixel stream decodes to the 320x200
    picture, its quarter-block output is what the oracle produces for the same pixels."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "abi_demo")
    if not os.path.exists(exe):
        pytest.skip("examples/abi_demo not built")
    six = subprocess.run([exe], capture_output=True, timeout=120)
    assert six.returncode == 0, six.stderr.decode()
    img, ncolors = oracle.sixel_decode(six.stdout[six.stdout.index(b"\x1bP"):])
    assert img.shape[:2] == (204, 320) and 2 <= ncolors <= 256
    q = subprocess.run([exe, "quarter"], capture_output=True, timeout=120)
    assert q.returncode == 0 and q.stdout.count(b"\n") == 100 and b"\xe2\x96" in q.stdout
