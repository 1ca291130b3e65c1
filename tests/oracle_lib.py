"""ctypes wrappers of the CHECKERS: oracle/libtimg_oracle.so (this repo's CPU
restatement) and oracle/_ref/libtimg_ref.so (the real reference, compiled from
/root/reference where available).  Test infrastructure only."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_int, c_long, c_size_t, c_uint32, c_void_p, POINTER

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vp = c_void_p


def pack(c):
    return int(c[0]) | int(c[1]) << 8 | int(c[2]) << 16 | int(c[3]) << 24


def _d(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"] and a.dtype == np.uint8
    return a.ctypes.data_as(vp)


class Oracle:
    def __init__(self):
        self.L = L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libtimg_oracle.so"))
        L.oracle_block_encode.restype = c_long
        L.oracle_block_max_bytes.restype = c_size_t
        L.oracle_block_canvas_new.restype = vp
        L.oracle_block_canvas_free.argtypes = [vp]
        L.oracle_block_canvas_send.restype = c_long
        L.oracle_block_canvas_send.argtypes = [vp, c_int, c_int, vp, c_int, c_int, vp, c_long]
        if not hasattr(L, "oracle_sixel_encode"):
            return
        L.oracle_sixel_encode.restype = c_long
        L.oracle_sixel_encode.argtypes = [vp, c_int, c_int, c_int, c_uint32, c_uint32, c_int,
                                          c_int, c_int, c_int, vp, c_long]
        L.oracle_sixel_decode.argtypes = [vp, c_long, vp, c_int, c_int, POINTER(c_int),
                                          POINTER(c_int), POINTER(c_int)]
        L.oracle_sixel_palette.argtypes = [vp, c_int, c_int, vp, POINTER(c_int)]

    def scale(self, src: np.ndarray, dw, dh, in_fmt=0, filter=0) -> np.ndarray:
        sh, sw = src.shape[:2]
        src = np.ascontiguousarray(src)
        dst = np.zeros((dh, dw, 4), np.uint8)
        rc = self.L.oracle_scale(_d(src), sw, sh, in_fmt, _d(dst), dw, dh, filter)
        assert rc == 0
        return dst

    def plan_info(self, sw, sh, dw, dh, filter=0):
        a = (c_int * 6)()
        assert self.L.oracle_scale_plan_info(sw, sh, dw, dh, filter, a) == 0
        return dict(zip(["vertical_first", "h_widest", "v_is_gather", "v_widest", "h_filter",
                         "v_filter"], list(a)))

    def plan_dump(self, sw, sh, dw, dh, filter=0):
        return _plan_dump(self.L.oracle_scale_plan_dump, (sw, sh), dw, dh, filter, sw, sh)

    def alpha_compose(self, fb: np.ndarray, bg, pattern=(0, 0, 0, 0), pw=0, ph=0, start_row=0,
                      has_getter=True):
        out = np.ascontiguousarray(fb).copy()
        h, w = out.shape[:2]
        calls = self.L.oracle_alpha_compose(_d(out), w, h, int(has_getter), c_uint32(pack(bg)),
                                            c_uint32(pack(pattern)), pw, ph, start_row)
        return out, calls

    def block_max_bytes(self, w, h):
        return int(self.L.oracle_block_max_bytes(w, h))

    def block_encode(self, fb: np.ndarray, quarter=False, upper=False, color256=False, x=0) -> bytes:
        fb = np.ascontiguousarray(fb)
        h, w = fb.shape[:2]
        cap = self.block_max_bytes(w, h) + 64
        out = ctypes.create_string_buffer(cap)
        n = self.L.oracle_block_encode(_d(fb), w, h, int(quarter), int(upper), int(color256), x,
                                       out, cap)
        assert n >= 0
        return out.raw[:n]

    def block_canvas(self, quarter=False, upper=False, color256=False):
        return _OracleCanvas(self, quarter, upper, color256)

    def as_256(self, c):
        return int(self.L.oracle_as_256_term_color(c_uint32(pack(c)))) & 0xFF

    # sixel
    def sixel_encode(self, fb, bg=(0, 0, 0, 0), pattern=(0, 0, 0, 0), pw=0, ph=0, has_getter=True,
                     broken_cursor=False, lookup_mode=1) -> bytes:
        fb = np.ascontiguousarray(fb)
        h, w = fb.shape[:2]
        cap = 4096 + w * (h + 6) * 8
        out = ctypes.create_string_buffer(cap)
        n = self.L.oracle_sixel_encode(_d(fb), w, h, int(has_getter), pack(bg), pack(pattern), pw,
                                       ph, int(broken_cursor), lookup_mode, out, cap)
        assert n >= 0, n
        return out.raw[:n]

    def sixel_decode(self, data: bytes, cap_w=4096, cap_h=4096):
        img = np.zeros((cap_h, cap_w, 4), np.uint8)
        w, h, nc = c_int(), c_int(), c_int()
        rc = self.L.oracle_sixel_decode(data, len(data), _d(img), cap_w, cap_h, w, h, nc)
        assert rc == 0, rc
        return img[:h.value, :w.value].copy(), nc.value

    def sixel_palette(self, fb):
        fb = np.ascontiguousarray(fb)
        h, w = fb.shape[:2]
        pal = np.zeros((256, 3), np.uint8)
        off = c_int()
        n = self.L.oracle_sixel_palette(_d(fb), w, h, _d(pal), off)
        return pal[:n].copy(), bool(off.value)


class _OracleCanvas:
    def __init__(self, o: Oracle, quarter, upper, color256):
        self.o = o
        self.h = vp(o.L.oracle_block_canvas_new(int(quarter), int(upper), int(color256)))

    def send(self, x, dy, fb: np.ndarray) -> bytes:
        fb = np.ascontiguousarray(fb)
        h, w = fb.shape[:2]
        cap = self.o.block_max_bytes(w, h) + 64
        out = ctypes.create_string_buffer(cap)
        n = self.o.L.oracle_block_canvas_send(self.h, x, dy, _d(fb), w, h, out, cap)
        assert n >= 0
        return out.raw[:n]

    def close(self):
        self.o.L.oracle_block_canvas_free(self.h)


def _plan_dump(fn, lead, dw, dh, filter, sw, sh):
    hdr = (c_int * 8)()
    ht = (c_int * (2 * dw))()
    hcap = dw * (int(np.ceil(4 * max(1.0, sw / dw))) + 16)
    hc = (ctypes.c_float * hcap)()
    vcap = dh * (int(np.ceil(4 * max(1.0, sh / dh))) + 16) + 64
    vc = (c_int * dh)()
    vr = (c_int * vcap)()
    vco = (ctypes.c_float * vcap)()
    n = fn(*lead, dw, dh, filter, hdr, ht, hc, hcap, vc, vr, vco, vcap)
    assert n >= 0
    hw = hdr[3]
    return dict(
        header=list(hdr)[:4],
        h_taps=np.array(ht[:]),
        h_coeff=np.frombuffer(hc, dtype=np.float32)[:dw * hw].copy().view(np.uint32),
        v_cnt=np.array(vc[:]),
        v_rows=np.array(vr[:n]),
        v_coeff=np.frombuffer(vco, dtype=np.float32)[:n].copy().view(np.uint32),
    )


def product_plan_dump(sw, sh, dw, dh, filter=0, in_fmt=0):
    """The product's independently built tables (host-only debug export)."""
    import timg_amd
    L = timg_amd.load_library()
    return _plan_dump(L.timg_hip_debug_plan_dump, (sw, sh, in_fmt), dw, dh, filter, sw, sh)


class Ref:
    """The real reference (hzeller/timg sources compiled into oracle/_ref)."""

    @staticmethod
    def try_load():
        p = os.path.join(ROOT, "oracle", "_ref", "libtimg_ref.so")
        if not os.path.exists(p):
            return None
        return Ref(p)

    def __init__(self, path):
        self.L = L = ctypes.CDLL(path)
        L.ref_block_encode.restype = c_long
        L.ref_block_canvas_new.restype = vp
        L.ref_block_canvas_send.argtypes = [vp, c_int, c_int, vp, c_int, c_int]
        L.ref_block_canvas_read.restype = c_long
        L.ref_block_canvas_read.argtypes = [vp, vp, c_long]
        L.ref_block_canvas_free.argtypes = [vp]

    def scale(self, src, dw, dh, in_fmt=0):
        sh, sw = src.shape[:2]
        src = np.ascontiguousarray(src)
        dst = np.zeros((dh, dw, 4), np.uint8)
        assert self.L.ref_scale(_d(src), sw, sh, in_fmt, _d(dst), dw, dh) == 0
        return dst

    def alpha_compose(self, fb, bg, pattern=(0, 0, 0, 0), pw=0, ph=0, start_row=0, has_getter=True):
        out = np.ascontiguousarray(fb).copy()
        h, w = out.shape[:2]
        calls = self.L.ref_alpha_compose(_d(out), w, h, int(has_getter), c_uint32(pack(bg)),
                                         c_uint32(pack(pattern)), pw, ph, start_row)
        return out, calls

    def block_encode(self, fb, quarter=False, upper=False, color256=False, x=0) -> bytes:
        fb = np.ascontiguousarray(fb)
        h, w = fb.shape[:2]
        cap = 64 + ((h + 1) // 2) * (16 + w * 40)
        out = ctypes.create_string_buffer(cap)
        n = self.L.ref_block_encode(_d(fb), w, h, int(quarter), int(upper), int(color256), x, out,
                                    cap)
        assert 0 <= n <= cap
        return out.raw[:n]

    def block_canvas(self, quarter=False, upper=False, color256=False):
        return _RefCanvas(self, quarter, upper, color256)

    def as_256(self, c):
        return int(self.L.ref_as_256_term_color(c_uint32(pack(c)))) & 0xFF


class _RefCanvas:
    def __init__(self, r: Ref, quarter, upper, color256):
        self.r = r
        self.h = vp(r.L.ref_block_canvas_new(int(quarter), int(upper), int(color256)))

    def send(self, x, dy, fb):
        fb = np.ascontiguousarray(fb)
        h, w = fb.shape[:2]
        self.r.L.ref_block_canvas_send(self.h, x, dy, _d(fb), w, h)

    def read_all(self) -> bytes:
        n = self.r.L.ref_block_canvas_read(self.h, None, 0)
        out = ctypes.create_string_buffer(max(n, 1))
        self.r.L.ref_block_canvas_read(self.h, out, n)
        return out.raw[:n]

    def close(self):
        self.r.L.ref_block_canvas_free(self.h)
