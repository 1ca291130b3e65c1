"""ctypes binding of include/timg_hip.h (tests / bench plumbing).

No torch types cross the boundary: callers hand over raw pointers (ints) for
device memory -- e.g. ``tensor.data_ptr()`` -- or numpy arrays for host memory.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, byref, c_char_p, c_int, c_size_t, c_uint32,
                    c_uint8, c_void_p)

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path() -> str:
    # TIMG_HIP_LIB: an experiment build of the same library (scratch/build_variant.sh)
    return os.environ.get("TIMG_HIP_LIB") or os.path.join(_HERE, "libtimg_hip.so")


class TimgHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"timg_hip error {code}: {msg}")
        self.code = code


class Blend(ctypes.Structure):
    """timg_hip_blend"""
    _fields_ = [("enabled", c_int), ("bg", c_uint32), ("pattern", c_uint32),
                ("pattern_w", c_int), ("pattern_h", c_int), ("start_row", c_int)]

    @staticmethod
    def make(bg, pattern=(0, 0, 0, 0), pw=0, ph=0, start_row=0, enabled=True):
        def pack(c):
            return int(c[0]) | int(c[1]) << 8 | int(c[2]) << 16 | int(c[3]) << 24
        return Blend(1 if enabled else 0, pack(bg), pack(pattern), pw, ph, start_row)


_lib = None


def load_library():
    """Loads libtimg_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch wheels bundle their own libamdhip64.so.7; the process can only hold
    # one HIP runtime (same SONAME), and torch refuses to run on a foreign one.
    # Let torch bring its runtime in first -- libtimg_hip.so is happy with either.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional plumbing
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            " (make -C timg_amd/csrc). There is no CPU fallback.")
    L = ctypes.CDLL(path)
    vp = c_void_p
    L.timg_hip_init.argtypes = [c_int, POINTER(vp)]
    L.timg_hip_destroy.argtypes = [vp]
    L.timg_hip_destroy.restype = None
    L.timg_hip_last_error.argtypes = [vp]
    L.timg_hip_last_error.restype = c_char_p
    L.timg_hip_malloc.argtypes = [vp, c_size_t, POINTER(vp)]
    L.timg_hip_free.argtypes = [vp, vp]
    L.timg_hip_memcpy_h2d.argtypes = [vp, vp, vp, c_size_t, vp]
    L.timg_hip_memcpy_d2h.argtypes = [vp, vp, vp, c_size_t, vp]
    L.timg_hip_sync.argtypes = [vp, vp]
    L.timg_hip_memcpy_d2d.argtypes = [vp, vp, vp, c_size_t, vp]
    L.timg_hip_stream_create.argtypes = [vp, c_int, c_int, POINTER(vp)]
    L.timg_hip_stream_destroy.argtypes = [vp, vp]
    L.timg_hip_stream_wait_stream.argtypes = [vp, vp, vp]
    L.timg_hip_synth_frames.argtypes = [vp, c_int, c_int, c_int, c_uint32, c_int, c_int, vp, c_size_t, c_int, vp]
    L.timg_hip_scaler_create.argtypes = [vp, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(vp)]
    L.timg_hip_scaler_destroy.argtypes = [vp]
    L.timg_hip_scaler_destroy.restype = None
    L.timg_hip_scaler_set_kernel.argtypes = [vp, c_int]
    L.timg_hip_scaler_info.argtypes = [vp, POINTER(c_int)]
    L.timg_hip_scaler_algorithmic_bytes.argtypes = [vp]
    L.timg_hip_scaler_algorithmic_bytes.restype = c_size_t
    L.timg_hip_scale_blend.argtypes = [vp, vp, vp, c_int, c_size_t, c_int, vp, c_int, c_size_t,
                                       c_int, c_int, POINTER(Blend), POINTER(c_int), vp]
    L.timg_hip_alpha_compose.argtypes = [vp, vp, c_int, c_int, c_int, c_size_t, c_int, c_int,
                                         POINTER(Blend), POINTER(c_int), vp]
    L.timg_hip_autocrop_bbox.argtypes = [vp, vp, c_int, c_int, c_int, c_size_t, c_int, c_int,
                                         c_int, POINTER(c_int), vp]
    L.timg_hip_block_max_bytes.argtypes = [c_int, c_int]
    L.timg_hip_block_max_bytes.restype = c_size_t
    L.timg_hip_block_encode.argtypes = [vp, vp, c_int, c_int, c_int, c_size_t, c_int, c_int,
                                        c_int, c_int, vp, c_size_t, c_int, POINTER(c_size_t), vp]
    L.timg_hip_block_encode_grid.argtypes = [vp, vp, c_int, c_int, c_int, c_size_t, c_int, c_int,
                                             c_int, POINTER(c_int), vp, c_size_t, c_int,
                                             POINTER(c_size_t), vp]
    L.timg_hip_block_canvas_create.argtypes = [vp, c_int, POINTER(vp)]
    L.timg_hip_block_canvas_destroy.argtypes = [vp]
    L.timg_hip_block_canvas_destroy.restype = None
    L.timg_hip_block_canvas_forget.argtypes = [vp]
    L.timg_hip_block_canvas_forget.restype = None
    L.timg_hip_block_canvas_send.argtypes = [vp, c_int, c_int, vp, c_int, c_int, c_int, c_int, vp,
                                             c_size_t, POINTER(c_size_t), vp]
    L.timg_hip_sixel_max_bytes.argtypes = [c_int, c_int]
    L.timg_hip_sixel_max_bytes.restype = c_size_t
    L.timg_hip_sixel_encode.argtypes = [vp, vp, c_int, c_int, c_int, c_size_t, c_int, c_int,
                                        c_int, POINTER(Blend), vp, c_size_t, c_int,
                                        POINTER(c_size_t), vp]
    L.timg_hip_sixel_job_create.argtypes = [vp, c_int, POINTER(vp)]
    L.timg_hip_sixel_job_destroy.argtypes = [vp]
    L.timg_hip_sixel_job_destroy.restype = None
    L.timg_hip_sixel_encode_async.argtypes = [vp, vp, c_int, c_int, c_int, c_size_t, c_int, c_int, POINTER(Blend), vp,
                                              c_size_t, vp, vp]
    L.timg_hip_sixel_encode_wait.argtypes = [vp, POINTER(c_size_t)]
    L.timg_hip_scale_sixel_encode.argtypes = [vp, vp, vp, c_int, c_size_t, vp, c_int, POINTER(Blend), c_int, vp, c_size_t,
                                              c_int, POINTER(c_size_t), c_int, POINTER(ctypes.c_float), vp]
    L.timg_hip_png_bytes.argtypes = [c_int, c_int, c_int]
    L.timg_hip_png_bytes.restype = c_size_t
    L.timg_hip_gfx_max_bytes.argtypes = [c_int, c_int]
    L.timg_hip_gfx_max_bytes.restype = c_size_t
    gfx = [vp, vp, c_int, c_int, c_int, c_size_t, c_int, c_int, c_int]
    L.timg_hip_png_encode.argtypes = gfx + [vp, c_size_t, c_int, POINTER(c_size_t), vp]
    L.timg_hip_iterm2_encode.argtypes = gfx + [vp, c_size_t, c_int, POINTER(c_size_t), vp]
    L.timg_hip_kitty_encode.argtypes = gfx + [vp, vp, c_size_t, c_int, POINTER(c_size_t), vp]
    _lib = L
    return L


def _ptr(x):
    """numpy array -> (pointer, False); int device pointer -> (pointer, True)."""
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return c_void_p(x.ctypes.data), False
    return c_void_p(int(x)), True


class Scaler:
    def __init__(self, owner: "TimgHip", handle, in_w, in_h, out_w, out_h):
        self.owner, self.handle = owner, handle
        self.in_w, self.in_h, self.out_w, self.out_h = in_w, in_h, out_w, out_h

    def info(self):
        a = (c_int * 8)()
        self.owner._check(self.owner.L.timg_hip_scaler_info(self.handle, a))
        keys = ["vertical_first", "h_widest", "v_is_gather", "v_widest", "h_filter",
                "v_filter", "streaming_ok", "max_active_rows"]
        d = dict(zip(keys, list(a)))
        bits = d["streaming_ok"]
        d.update(streaming_ok=bits & 1, matrix_kernel=(bits >> 1) & 1, matrix_overflow_row=(bits >> 2) & 1,
                 two_column_kernel=(bits >> 3) & 1)
        return d

    def set_kernel(self, which: int):
        self.owner._check(self.owner.L.timg_hip_scaler_set_kernel(self.handle, which))

    def algorithmic_bytes(self) -> int:
        return int(self.owner.L.timg_hip_scaler_algorithmic_bytes(self.handle))

    def close(self):
        if self.handle:
            self.owner.L.timg_hip_scaler_destroy(self.handle)
            self.handle = None


class BlockCanvas:
    """timg_hip_block_canvas: one UnicodeBlockCanvas' worth of state (frame-diff)."""

    def __init__(self, owner: "TimgHip", flags: int):
        self.owner = owner
        h = c_void_p()
        owner._check(owner.L.timg_hip_block_canvas_create(owner.ctx, flags, byref(h)))
        self.handle = h

    def send(self, x: int, dy: int, fb, w: int, h: int, stride: int = 0) -> bytes:
        """Bytes Send(x, dy, fb) appends after its cursor prefix."""
        p, dev = _ptr(fb)
        cap = self.owner.block_max_bytes(w, h)
        out = np.empty(cap, np.uint8)
        n = c_size_t()
        self.owner._check(self.owner.L.timg_hip_block_canvas_send(
            self.handle, x, dy, p, w, h, stride, int(dev), c_void_p(out.ctypes.data), cap,
            byref(n), None))
        return out[:n.value].tobytes()

    def forget(self):
        """The next send encodes every cell (as after a Send at another position)."""
        self.owner.L.timg_hip_block_canvas_forget(self.handle)

    def close(self):
        if self.handle:
            self.owner.L.timg_hip_block_canvas_destroy(self.handle)
            self.handle = None


class TimgHip:
    """One timg_hip_ctx."""

    QUARTER, UPPER, COLOR256 = 1, 2, 4
    SIXEL_BROKEN_CURSOR = 1  # timg_hip_sixel_encode flag

    def __init__(self, device: int = 0):
        self.L = load_library()
        h = c_void_p()
        rc = self.L.timg_hip_init(device, byref(h))
        if rc != 0:
            raise TimgHipError(rc, (self.L.timg_hip_last_error(None) or b"").decode())
        self.ctx = h

    def close(self):
        if self.ctx:
            self.L.timg_hip_destroy(self.ctx)
            self.ctx = None

    def _check(self, rc):
        if rc != 0:
            raise TimgHipError(rc, (self.L.timg_hip_last_error(self.ctx) or b"").decode())

    # -- memory helpers ------------------------------------------------------
    def malloc(self, nbytes: int) -> int:
        p = c_void_p()
        self._check(self.L.timg_hip_malloc(self.ctx, nbytes, byref(p)))
        return p.value

    def free(self, ptr: int):
        self._check(self.L.timg_hip_free(self.ctx, c_void_p(ptr)))

    def upload(self, arr: np.ndarray, dptr: int | None = None) -> int:
        arr = np.ascontiguousarray(arr)
        if dptr is None:
            dptr = self.malloc(arr.nbytes)
        self._check(self.L.timg_hip_memcpy_h2d(self.ctx, c_void_p(dptr), c_void_p(arr.ctypes.data),
                                               arr.nbytes, None))
        return dptr

    def download(self, dptr: int, nbytes: int) -> np.ndarray:
        out = np.empty(nbytes, np.uint8)
        self._check(self.L.timg_hip_memcpy_d2h(self.ctx, c_void_p(out.ctypes.data), c_void_p(dptr),
                                               nbytes, None))
        return out

    def sync(self, stream=None):
        self._check(self.L.timg_hip_sync(self.ctx, c_void_p(stream) if stream else None))

    def stream_create(self, reserved_cus_per_xcd: int = 0, high_priority: bool = False) -> int:
        """A stream of the context (timg_hip_stream_create): its kernels stay off `reserved_cus_per_xcd` CUs of every XCD
        (a multiple of 4), or it has the device's greatest priority.  Returns the raw hipStream_t."""
        st = c_void_p()
        self._check(self.L.timg_hip_stream_create(self.ctx, reserved_cus_per_xcd, 1 if high_priority else 0, byref(st)))
        return st.value

    def stream_destroy(self, stream: int):
        self._check(self.L.timg_hip_stream_destroy(self.ctx, c_void_p(stream)))

    def stream_wait_stream(self, waiter: int, signaller: int):
        """`waiter`'s later work waits for everything `signaller` holds now (device-side; nobody blocks)."""
        self._check(self.L.timg_hip_stream_wait_stream(self.ctx, c_void_p(waiter), c_void_p(signaller)))

    def synth_frames(self, kind: str, w: int, h: int, seed: int = 0, first_frame: int = 0, n_frames: int = 1,
                     dst: int | None = None, frame_stride: int = 0, stream=None):
        """The benchmark's synthetic frames (timg_hip_synth_frames).  dst: a device pointer (frames
        are written there) or None (frames come back as a numpy array)."""
        from .synth import HASH_KINDS
        if dst is not None:
            self._check(self.L.timg_hip_synth_frames(self.ctx, HASH_KINDS[kind], w, h, seed, first_frame, n_frames,
                                                     c_void_p(dst), frame_stride, 1,
                                                     c_void_p(stream) if stream else None))
            return None
        out = np.empty((n_frames, h, w, 4), np.uint8)
        self._check(self.L.timg_hip_synth_frames(self.ctx, HASH_KINDS[kind], w, h, seed, first_frame, n_frames,
                                                 c_void_p(out.ctypes.data), 0, 0, None))
        return out

    # -- scaler --------------------------------------------------------------
    def scaler(self, in_w, in_h, out_w, out_h, in_fmt=0, filter=0) -> Scaler:
        h = c_void_p()
        self._check(self.L.timg_hip_scaler_create(self.ctx, in_w, in_h, in_fmt, out_w, out_h,
                                                  filter, byref(h)))
        return Scaler(self, h, in_w, in_h, out_w, out_h)

    def scale_blend(self, scaler: Scaler, src, dst, n_frames=1, blend: Blend | None = None,
                    want_transparent=False, stream=None, src_stride=0, dst_stride=0,
                    src_frame_stride=0, dst_frame_stride=0):
        sp, s_dev = _ptr(src)
        dp, d_dev = _ptr(dst)
        flags = (c_int * n_frames)() if want_transparent else None
        self._check(self.L.timg_hip_scale_blend(
            self.ctx, scaler.handle, sp, src_stride, src_frame_stride, int(s_dev), dp, dst_stride,
            dst_frame_stride, int(d_dev), n_frames, byref(blend) if blend is not None else None,
            flags, c_void_p(stream) if stream else None))
        return list(flags) if want_transparent else None

    def scale(self, src: np.ndarray, out_w, out_h, in_fmt=0, filter=0, blend=None,
              kernel=0) -> np.ndarray:
        """Host convenience: (H,W,4) uint8 -> (out_h,out_w,4) uint8."""
        sh, sw = src.shape[:2]
        sc = self.scaler(sw, sh, out_w, out_h, in_fmt, filter)
        try:
            if kernel:
                sc.set_kernel(kernel)
            dst = np.empty((out_h, out_w, 4), np.uint8)
            self.scale_blend(sc, np.ascontiguousarray(src), dst, 1, blend)
            return dst
        finally:
            sc.close()

    def alpha_compose(self, fb, w, h, blend: Blend, n_frames=1, want_transparent=False,
                      stride=0, frame_stride=0, stream=None):
        p, dev = _ptr(fb)
        flags = (c_int * n_frames)() if want_transparent else None
        self._check(self.L.timg_hip_alpha_compose(self.ctx, p, w, h, stride, frame_stride,
                                                  int(dev), n_frames, byref(blend), flags,
                                                  c_void_p(stream) if stream else None))
        return list(flags) if want_transparent else None

    def autocrop_bbox(self, src, w, h, n_frames=1, crop_border=0, stride=0, frame_stride=0):
        p, dev = _ptr(src)
        out = (c_int * (4 * n_frames))()
        self._check(self.L.timg_hip_autocrop_bbox(self.ctx, p, w, h, stride, frame_stride,
                                                  int(dev), n_frames, crop_border, out, None))
        return np.array(out[:]).reshape(n_frames, 4)

    # -- canvases --------------------------------------------------------------
    def block_max_bytes(self, w, h) -> int:
        return int(self.L.timg_hip_block_max_bytes(w, h))

    def block_encode(self, fb, w, h, flags=0, x_indent=0, n_frames=1, out=None, out_cap=None,
                     stride=0, frame_stride=0, stream=None, x_indents=None):
        """Returns list[bytes] (host out) or the lengths (device out).  x_indents: one Send x
        per frame (a grid row) instead of the common x_indent."""
        p, dev = _ptr(fb)
        if out_cap is None:
            out_cap = self.block_max_bytes(w, h)
        host_out = out is None
        if host_out:
            out = np.empty(out_cap * n_frames, np.uint8)
        op, o_dev = _ptr(out)
        lens = (c_size_t * n_frames)()
        if x_indents is not None:
            xs = (c_int * n_frames)(*[int(v) for v in x_indents])
            self._check(self.L.timg_hip_block_encode_grid(self.ctx, p, w, h, stride, frame_stride, int(dev),
                                                          n_frames, flags, xs, op, out_cap, int(o_dev), lens,
                                                          c_void_p(stream) if stream else None))
        else:
            self._check(self.L.timg_hip_block_encode(self.ctx, p, w, h, stride, frame_stride, int(dev),
                                                     n_frames, flags, x_indent, op, out_cap,
                                                     int(o_dev), lens,
                                                     c_void_p(stream) if stream else None))
        if host_out:
            return [out[i * out_cap:i * out_cap + lens[i]].tobytes() for i in range(n_frames)]
        return list(lens)

    def block_canvas(self, flags=0) -> BlockCanvas:
        return BlockCanvas(self, flags)

    # ---- graphics protocols at --compress=0 (png / kitty / iTerm2) ----
    RGB24 = 1

    def gfx_encode(self, kind, fb, w, h, n_frames=1, rgb24=False, image_ids=None, out=None, out_cap=None,
                   stride=0, frame_stride=0, stream=None):
        """kind: "png", "kitty" (image_ids: one uint32 per frame) or "iterm2"."""
        p, dev = _ptr(fb)
        if out_cap is None:
            out_cap = int(self.L.timg_hip_gfx_max_bytes(w, h))
        host_out = out is None
        if host_out:
            out = np.empty(out_cap * n_frames, np.uint8)
        op, o_dev = _ptr(out)
        lens = (c_size_t * n_frames)()
        flags = self.RGB24 if rgb24 else 0
        st = c_void_p(stream) if stream else None
        if kind == "kitty":
            ids = (ctypes.c_uint32 * n_frames)(*[int(v) for v in image_ids])
            rc = self.L.timg_hip_kitty_encode(self.ctx, p, w, h, stride, frame_stride, int(dev), n_frames, flags,
                                              ids, op, out_cap, int(o_dev), lens, st)
        else:
            fn = self.L.timg_hip_png_encode if kind == "png" else self.L.timg_hip_iterm2_encode
            rc = fn(self.ctx, p, w, h, stride, frame_stride, int(dev), n_frames, flags, op, out_cap, int(o_dev),
                    lens, st)
        self._check(rc)
        if host_out:
            return [out[i * out_cap:i * out_cap + lens[i]].tobytes() for i in range(n_frames)]
        return list(lens)

    def scale_sixel_encode(self, scaler: Scaler, src: int, scaled: int, n_frames: int, blend: Blend | None, out: int,
                           out_cap: int, flags=0, pieces=0, stream=None, src_stride=0, src_frame_stride=0):
        """timg_hip_scale_sixel_encode on device memory (src, scaled, out: device pointers).  Returns
        (lengths, device ms of the scale kernels summed over the pieces)."""
        lens = (c_size_t * n_frames)()
        ms = ctypes.c_float(0.0)
        self._check(self.L.timg_hip_scale_sixel_encode(
            self.ctx, scaler.handle, c_void_p(int(src)), src_stride, src_frame_stride, c_void_p(int(scaled)), n_frames,
            byref(blend) if blend is not None else None, flags, c_void_p(int(out)), out_cap, 1, lens, pieces, byref(ms),
            c_void_p(stream) if stream else None))
        return list(lens), float(ms.value)

    def sixel_max_bytes(self, w, h) -> int:
        return int(self.L.timg_hip_sixel_max_bytes(w, h))

    def sixel_encode_first_hit(self, fb, w, h, flags=0, pad_blend: Blend | None = None, n_frames=1, out_cap=None):
        """TEST-ONLY: the encoder under libsixel's own first-hit lookup rule, from libtimg_hip_debug.so
        (timg_hip_debug_sixel_encode_first_hit: serial, ~0.3 s per 800x450 frame); host output."""
        if not hasattr(self, "_debug_lib"):
            self._debug_lib = ctypes.CDLL(os.path.join(_HERE, "libtimg_hip_debug.so"))
            self._debug_lib.timg_hip_debug_sixel_encode_first_hit.restype = ctypes.c_int
        p, dev = _ptr(fb)
        if out_cap is None:
            out_cap = self.sixel_max_bytes(w, h)
        out = np.empty(out_cap * n_frames, np.uint8)
        lens = (c_size_t * n_frames)()
        self._check(self._debug_lib.timg_hip_debug_sixel_encode_first_hit(
            self.ctx, p, c_int(w), c_int(h), c_int(0), c_size_t(0), c_int(int(dev)), c_int(n_frames), c_int(flags),
            byref(pad_blend) if pad_blend is not None else None, c_void_p(out.ctypes.data), c_size_t(out_cap), c_int(0),
            lens, None))
        return [out[i * out_cap:i * out_cap + lens[i]].tobytes() for i in range(n_frames)]

    # -- the asynchronous form: enqueue now, read the byte counts later (timg_hip_sixel_encode_async) --
    def sixel_job(self, max_frames: int):
        j = c_void_p()
        self._check(self.L.timg_hip_sixel_job_create(self.ctx, max_frames, byref(j)))
        return j

    def sixel_job_destroy(self, job):
        self.L.timg_hip_sixel_job_destroy(job)

    def sixel_encode_async(self, job, fb_dev: int, w, h, out_dev: int, out_cap, n_frames=1, flags=0,
                           pad_blend: Blend | None = None, stride=0, frame_stride=0, stream=None):
        self._check(self.L.timg_hip_sixel_encode_async(self.ctx, c_void_p(int(fb_dev)), w, h, stride, frame_stride, n_frames,
                                                       flags, byref(pad_blend) if pad_blend is not None else None,
                                                       c_void_p(int(out_dev)), out_cap,
                                                       c_void_p(stream) if stream else None, job))

    def sixel_encode_wait(self, job, n_frames: int):
        lens = (c_size_t * n_frames)()
        self._check(self.L.timg_hip_sixel_encode_wait(job, lens))
        return list(lens)

    def sixel_encode(self, fb, w, h, flags=0, pad_blend: Blend | None = None, n_frames=1,
                     out=None, out_cap=None, stride=0, frame_stride=0, stream=None):
        p, dev = _ptr(fb)
        if out_cap is None:
            out_cap = self.sixel_max_bytes(w, h)
        host_out = out is None
        if host_out:
            out = np.empty(out_cap * n_frames, np.uint8)
        op, o_dev = _ptr(out)
        lens = (c_size_t * n_frames)()
        self._check(self.L.timg_hip_sixel_encode(self.ctx, p, w, h, stride, frame_stride, int(dev),
                                                 n_frames, flags,
                                                 byref(pad_blend) if pad_blend is not None else None,
                                                 op, out_cap, int(o_dev), lens,
                                                 c_void_p(stream) if stream else None))
        if host_out:
            return [out[i * out_cap:i * out_cap + lens[i]].tobytes() for i in range(n_frames)]
        return list(lens)
