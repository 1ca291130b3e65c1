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

    def sixel_quantize_trace(self, fb, lookup_mode):
        """(palette[n,3], index[h,w], looked_up[h,w,3], dithered)"""
        fb = np.ascontiguousarray(fb)
        h, w = fb.shape[:2]
        pal = np.zeros((256, 3), np.uint8)
        idx = np.zeros((h, w), np.uint8)
        val = np.zeros((h, w, 3), np.uint8)
        d = c_int(0)
        n = self.L.oracle_sixel_quantize_trace(_d(fb), w, h, lookup_mode, _d(pal), _d(idx), _d(val), ctypes.byref(d))
        return pal[:n], idx, val, bool(d.value)

    def autocrop_bbox(self, fb, crop_border=0):
        fb = np.ascontiguousarray(fb)
        h, w = fb.shape[:2]
        out = (ctypes.c_int * 4)()
        self.L.oracle_autocrop_bbox(_d(fb), w, h, w * 4, crop_border, out)
        return list(out)

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

    # ---- graphics protocols at --compress=0 (oracle/png.c) ----
    def _png_setup(self):
        L = self.L
        if getattr(self, "_png_ready", False):
            return
        L.oracle_png_bytes.restype = c_size_t
        L.oracle_png_encode.restype = c_long
        L.oracle_png_encode.argtypes = [vp, c_int, c_int, c_int, vp, c_long]
        L.oracle_base64.restype = c_long
        L.oracle_base64.argtypes = [vp, c_long, vp]
        L.oracle_kitty_max_bytes.restype = c_size_t
        L.oracle_kitty_encode.restype = c_long
        L.oracle_kitty_encode.argtypes = [vp, c_int, c_int, c_int, c_uint32, vp, vp, c_long]
        L.oracle_iterm2_encode.restype = c_long
        L.oracle_iterm2_encode.argtypes = [vp, c_int, c_int, c_int, vp, vp, c_long]
        L.oracle_crc32.restype = c_uint32
        L.oracle_crc32.argtypes = [c_uint32, vp, c_size_t]
        L.oracle_adler32.restype = c_uint32
        L.oracle_adler32.argtypes = [vp, c_size_t]
        L.oracle_crc32_combine.restype = c_uint32
        L.oracle_crc32_combine.argtypes = [c_uint32, c_uint32, ctypes.c_uint64]
        self._png_ready = True

    def png_encode(self, fb, with_alpha=True) -> bytes:
        self._png_setup()
        fb = np.ascontiguousarray(fb)
        h, w = fb.shape[:2]
        cap = self.L.oracle_png_bytes(w, h, int(with_alpha))
        out = ctypes.create_string_buffer(cap)
        n = self.L.oracle_png_encode(_d(fb), w, h, int(with_alpha), out, cap)
        assert n == cap, (n, cap)
        return out.raw[:n]

    def base64(self, data: bytes) -> bytes:
        self._png_setup()
        out = ctypes.create_string_buffer((len(data) + 2) // 3 * 4 + 4)
        n = self.L.oracle_base64(ctypes.c_char_p(data), len(data), out)
        return out.raw[:n]

    def _gfx(self, fn, fb, with_alpha, *mid):
        self._png_setup()
        fb = np.ascontiguousarray(fb)
        h, w = fb.shape[:2]
        scratch = ctypes.create_string_buffer(self.L.oracle_png_bytes(w, h, int(with_alpha)))
        cap = self.L.oracle_kitty_max_bytes(w, h)
        out = ctypes.create_string_buffer(cap)
        n = fn(_d(fb), w, h, int(with_alpha), *mid, scratch, out, cap)
        assert n > 0
        return out.raw[:n]

    def kitty_encode(self, fb, image_id: int, with_alpha=True) -> bytes:
        self._png_setup()
        return self._gfx(self.L.oracle_kitty_encode, fb, with_alpha, c_uint32(image_id))

    def iterm2_encode(self, fb, with_alpha=True) -> bytes:
        self._png_setup()
        return self._gfx(self.L.oracle_iterm2_encode, fb, with_alpha)

    def crc32(self, data: bytes, crc=0) -> int:
        self._png_setup()
        return int(self.L.oracle_crc32(crc, ctypes.c_char_p(data), len(data)))

    def adler32(self, data: bytes) -> int:
        self._png_setup()
        return int(self.L.oracle_adler32(ctypes.c_char_p(data), len(data)))

    def crc32_combine(self, a, b, len_b) -> int:
        self._png_setup()
        return int(self.L.oracle_crc32_combine(a, b, len_b))

    def sixel_decode(self, data: bytes, cap_w=4096, cap_h=4096):
        img = np.zeros((cap_h, cap_w, 4), np.uint8)
        w, h, nc = c_int(), c_int(), c_int()
        rc = self.L.oracle_sixel_decode(data, len(data), _d(img), cap_w, cap_h, w, h, nc)
        assert rc == 0, rc
        return img[:h.value, :w.value].copy(), nc.value

    def sixel_set_tie_order(self, mode: int):
        """0: the pinned (stable) order among equal sort keys; bits 1 / 2 reverse it for colours / boxes."""
        self.L.oracle_sixel_set_tie_order.argtypes = [c_int]
        self.L.oracle_sixel_set_tie_order.restype = None
        self.L.oracle_sixel_set_tie_order(mode)

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
    # (a test-only library next to the product: libtimg_hip.so itself carries no debug entry points)
    L = ctypes.CDLL(os.path.join(ROOT, "timg_amd", "libtimg_hip_debug.so"))
    return _plan_dump(L.timg_hip_debug_plan_dump, (sw, sh, in_fmt), dw, dh, filter, sw, sh)


class Ref:
    """The real reference (hzeller/timg sources compiled into oracle/_ref)."""

    @staticmethod
    def try_load():
        p = os.path.join(ROOT, "oracle", "_ref", "libtimg_ref.so")
        if not os.path.exists(p):
            return None
        try:
            return Ref(p)
        except OSError:  # e.g. a box without the libdeflate the library was linked against
            return None

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

    # ---- graphics protocols (only when the reference library was built with libdeflate) ----
    def has_png(self):
        return hasattr(self.L, "ref_png_encode")

    def png_encode(self, fb, level=0, with_alpha=True) -> bytes:
        fb = np.ascontiguousarray(fb)
        h, w = fb.shape[:2]
        self.L.ref_png_upper_bound.restype = c_size_t
        self.L.ref_png_encode.restype = c_long
        cap = self.L.ref_png_upper_bound(w, h)
        out = ctypes.create_string_buffer(cap)
        n = self.L.ref_png_encode(_d(fb), w, h, level, int(with_alpha), out, c_long(cap))
        assert n > 0
        return out.raw[:n]

    def graphics_send(self, kind, fb, level=0, local_alpha=False) -> bytes:
        """Everything one Send(0, 0, fb) of the real KittyGraphicsCanvas (kind 0) or
        ITerm2GraphicsCanvas (kind 1) writes to the terminal."""
        fb = np.ascontiguousarray(fb)
        h, w = fb.shape[:2]
        self.L.ref_graphics_send.restype = c_long
        cap = 4096 + (w * h * 4 + h + 1024) * 2
        out = ctypes.create_string_buffer(cap)
        n = self.L.ref_graphics_send(kind, _d(fb), w, h, level, int(local_alpha), out, c_long(cap))
        assert n > 0
        return out.raw[:n]

    def as_256(self, c):
        return int(self.L.ref_as_256_term_color(c_uint32(pack(c)))) & 0xFF

    # ---- the real timg::SixelCanvas over oracle/stub/sixel.h (libsixel calls -> oracle/sixel.c) ----
    def has_sixel(self):
        return hasattr(self.L, "ref_sixel_send")

    def sixel_send(self, fb, x=0, n_sends=1, cell_x_px=9, cell_y_px=18, bg=(0, 0, 0, 0), pattern=(0, 0, 0, 0),
                   pattern_size=1, has_getter=True, broken_cursor=False, full_cell_jump=False,
                   lookup_mode=1) -> bytes:
        """Everything n_sends Sends of the real SixelCanvas write to the terminal: the first at
        (x, 0), the others at (x, -height)."""
        fb = np.ascontiguousarray(fb)
        h, w = fb.shape[:2]
        self.L.ref_sixel_send.restype = c_long
        cap = (4096 + w * (h + 6) * 8) * n_sends
        out = ctypes.create_string_buffer(cap)
        n = self.L.ref_sixel_send(_d(fb), w, h, x, n_sends, cell_x_px, cell_y_px, int(has_getter),
                                  c_uint32(pack(bg)), c_uint32(pack(pattern)), pattern_size, int(broken_cursor),
                                  int(full_cell_jump), lookup_mode, out, c_long(cap))
        assert n > 0, n
        return out.raw[:n]

    def sixel_cell_height(self, pixels, cell_y_px, full_cell_jump=False) -> int:
        return int(self.L.ref_sixel_cell_height(pixels, cell_y_px, int(full_cell_jump)))


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
