"""ctypes binding of include/timg_hip_comm.h (libtimg_hip_comm.so): the RCCL gather of the encoded
frames to rank 0 behind its C-ABI.  Test / bench plumbing; the C++ caller is
timg_amd/twins/hip-gather-writer.cc."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_int, c_size_t, c_uint64, c_uint8, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ID_BYTES = 128
_lib = None


def comm_lib_path() -> str:
    return os.path.join(_HERE, "libtimg_hip_comm.so")


def load_comm_library():
    global _lib
    if _lib is not None:
        return _lib
    # the library binds the RCCL the process already maps (comm.hip): a process that is going to hold torch
    # must have torch's copy mapped BEFORE the first call, or it would end up with two RCCLs
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional plumbing
        pass
    L = ctypes.CDLL(comm_lib_path())
    vp = c_void_p
    L.timg_hip_comm_unique_id.argtypes = [vp]
    L.timg_hip_comm_create.argtypes = [c_int, c_int, c_int, vp, POINTER(vp)]
    L.timg_hip_comm_destroy.argtypes = [vp]
    L.timg_hip_comm_destroy.restype = None
    L.timg_hip_comm_last_error.argtypes = [vp]
    L.timg_hip_comm_last_error.restype = c_char_p
    L.timg_hip_gather_to_root.argtypes = [vp, c_int, vp, vp, c_int, c_int, vp, vp, c_size_t, POINTER(c_size_t), vp]
    L.timg_hip_gather_lengths.argtypes = [vp, vp, c_int, c_int, vp, vp]
    L.timg_hip_gather_payload.argtypes = [vp, c_int, vp, vp, c_int, vp, c_size_t, POINTER(c_size_t), vp]
    L.timg_hip_comm_rccl_info.argtypes = [ctypes.c_char_p, c_size_t, POINTER(c_int)]
    L.timg_hip_shard_locate.argtypes = [c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]
    L.timg_hip_shard_locate.restype = None
    L.timg_hip_shard_count.argtypes = [c_int, c_int, c_int, c_int]
    _lib = L
    return L


class NoRccl(RuntimeError):
    """No usable librccl in this process or on the loader's search path."""


def rccl_info():
    """(path of the RCCL that serves libtimg_hip_comm.so + how it was found, ncclGetVersion number)."""
    L = load_comm_library()
    buf = ctypes.create_string_buffer(1024)
    ver = c_int(0)
    rc = L.timg_hip_comm_rccl_info(buf, len(buf), byref(ver))
    if rc != 0:
        raise NoRccl((L.timg_hip_comm_last_error(None) or b"").decode())
    return buf.value.decode(), ver.value


def shard_locate(n_total, world, round_robin, frame):
    L = load_comm_library()
    r, i = c_int(), c_int()
    L.timg_hip_shard_locate(n_total, world, int(round_robin), frame, byref(r), byref(i))
    return r.value, i.value


def shard_count(n_total, world, round_robin, rank):
    return load_comm_library().timg_hip_shard_count(n_total, world, int(round_robin), rank)


# include/timg_hip_comm.h TIMG_HIP_COMM_PAYLOAD_READY: "the caller has waited for the payload's producer itself" -- the exchange
# then runs on the communicator's stream beside the caller's other streams (pass as `stream`)
PAYLOAD_READY = 1


class Comm:
    """One communicator per process (one process per GPU)."""

    def __init__(self, device: int, world: int, rank: int, unique_id: bytes):
        self.L = load_comm_library()
        self.world, self.rank = world, rank
        h = c_void_p()
        buf = (c_uint8 * ID_BYTES).from_buffer_copy(unique_id)
        rc = self.L.timg_hip_comm_create(device, world, rank, buf, byref(h))
        if rc != 0:
            raise (NoRccl if rc == -2 else RuntimeError)(
                "timg_hip_comm_create: " + (self.L.timg_hip_comm_last_error(None) or b"").decode())
        self.h = h

    @staticmethod
    def unique_id() -> bytes:
        L = load_comm_library()
        buf = (c_uint8 * ID_BYTES)()
        rc = L.timg_hip_comm_unique_id(buf)
        if rc != 0:
            raise (NoRccl if rc == -2 else RuntimeError)(
                "timg_hip_comm_unique_id: " + (L.timg_hip_comm_last_error(None) or b"").decode())
        return bytes(buf)

    def gather_to_root(self, payload_ptr: int, lengths, n_frames_max: int, recv_ptr: int = 0, recv_cap: int = 0,
                       root: int = 0, stream=None):
        """payload_ptr / recv_ptr: device pointers.  Returns (all_lengths[world, n_frames_max], total bytes)
        on the root, None elsewhere."""
        lens = np.ascontiguousarray(lengths, dtype=np.uint64)
        all_len = np.zeros((self.world, n_frames_max), np.uint64)
        got = c_size_t(0)
        rc = self.L.timg_hip_gather_to_root(self.h, root, c_void_p(payload_ptr), c_void_p(lens.ctypes.data), len(lens),
                                            n_frames_max, c_void_p(all_len.ctypes.data), c_void_p(recv_ptr), recv_cap,
                                            byref(got), c_void_p(stream) if stream else None)
        if rc != 0:
            raise RuntimeError("timg_hip_gather_to_root: " + (self.L.timg_hip_comm_last_error(self.h) or b"").decode())
        return (all_len, got.value) if self.rank == root else None

    def gather_lengths(self, lengths, n_frames_max: int, stream=None) -> np.ndarray:
        """Step 1 alone: all_lengths[world, n_frames_max] on every rank."""
        lens = np.ascontiguousarray(lengths, dtype=np.uint64)
        all_len = np.zeros((self.world, n_frames_max), np.uint64)
        rc = self.L.timg_hip_gather_lengths(self.h, c_void_p(lens.ctypes.data), len(lens), n_frames_max,
                                            c_void_p(all_len.ctypes.data), c_void_p(stream) if stream else None)
        if rc != 0:
            raise RuntimeError("timg_hip_gather_lengths: " + (self.L.timg_hip_comm_last_error(self.h) or b"").decode())
        return all_len

    def gather_payload(self, payload_ptr: int, all_lengths: np.ndarray, recv_ptr: int = 0, recv_cap: int = 0,
                       root: int = 0, stream=None) -> int:
        """Step 2 alone: returns the total on the root (0 elsewhere); raises with code -3 on every rank when
        the root's buffer is too small."""
        all_len = np.ascontiguousarray(all_lengths, dtype=np.uint64)
        got = c_size_t(0)
        rc = self.L.timg_hip_gather_payload(self.h, root, c_void_p(payload_ptr), c_void_p(all_len.ctypes.data),
                                            all_len.shape[1], c_void_p(recv_ptr), recv_cap, byref(got),
                                            c_void_p(stream) if stream else None)
        if rc != 0:
            e = RuntimeError("timg_hip_gather_payload: " + (self.L.timg_hip_comm_last_error(self.h) or b"").decode())
            e.code = rc
            raise e
        return got.value

    def close(self):
        if self.h:
            self.L.timg_hip_comm_destroy(self.h)
            self.h = None
