"""Seeded synthetic RGBA8 frames: S-noise, S-photo, S-alpha.

Pure numpy, deterministic for (kind, seed, w, h) -- used by the parity tests,
the golden-vector generator and bench.py.  bench.py uploads frames generated
here (or replicates them on the device); nothing crosses PCIe in the timed
region.
"""
from __future__ import annotations

import numpy as np

BASE_SEED = 0x71170000


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(BASE_SEED + int(seed)))


def noise(w: int, h: int, seed: int = 0, opaque: bool = False) -> np.ndarray:
    """i.i.d. uniform bytes: worst case for colour elision / RLE."""
    a = _rng(seed).integers(0, 256, (h, w, 4), dtype=np.uint8)
    if opaque:
        a[..., 3] = 255
    return a


def photo(w: int, h: int, seed: int = 0) -> np.ndarray:
    """Sum of three low-frequency 2-D sinusoids per channel + 4 % noise, opaque."""
    rng = _rng(seed)
    y = np.arange(h, dtype=np.float32)[:, None] / max(h, 1)
    x = np.arange(w, dtype=np.float32)[None, :] / max(w, 1)
    out = np.empty((h, w, 4), np.uint8)
    for c in range(3):
        acc = np.zeros((h, w), np.float32)
        for _ in range(3):
            fx, fy = rng.uniform(0.5, 6.0, 2)
            ph = rng.uniform(0, 2 * np.pi)
            acc += np.sin(2 * np.pi * (fx * x + fy * y) + ph).astype(np.float32)
        v = (acc / 6.0 + 0.5) * 255.0
        v += rng.normal(0.0, 0.04 * 255.0, (h, w)).astype(np.float32)
        out[..., c] = np.clip(v, 0, 255).astype(np.uint8)
    out[..., 3] = 255
    return out


def alpha(w: int, h: int, seed: int = 0) -> np.ndarray:
    """S-photo colours; A = radial ramp, 64-px transparent border (scaled for
    small frames), 10 % of pixels at exactly 0 / 0x5f / 0x60 / 0xff."""
    rng = _rng(seed + 7919)
    out = photo(w, h, seed)
    yy = (np.arange(h, dtype=np.float32)[:, None] - h / 2) / max(h / 2, 1)
    xx = (np.arange(w, dtype=np.float32)[None, :] - w / 2) / max(w / 2, 1)
    ramp = np.clip(1.0 - np.sqrt(xx * xx + yy * yy) / 1.2, 0, 1) * 255.0
    a = ramp.astype(np.uint8)
    b = max(1, min(64, min(w, h) // 8))
    a[:b, :] = 0
    a[-b:, :] = 0
    a[:, :b] = 0
    a[:, -b:] = 0
    pick = rng.random((h, w)) < 0.10
    special = rng.choice(np.array([0, 0x5F, 0x60, 0xFF], np.uint8), size=(h, w))
    a = np.where(pick, special, a)
    out[..., 3] = a
    return out


KINDS = {"noise": noise, "photo": photo, "alpha": alpha}


def make(kind: str, w: int, h: int, seed: int = 0) -> np.ndarray:
    return KINDS[kind](w, h, seed)


# ---- the benchmark's frames: counter-based integer hash, identical on host and device -------------
# (timg_amd/csrc/synth.hip: timg_hip_synth_frames is the same function of (kind, seed, frame, x, y))
HASH_KINDS = {"noise": 0, "photo": 1, "alpha": 2}
_M32 = np.uint64(0xFFFFFFFF)


def _mix(x):
    x = np.asarray(x, dtype=np.uint64) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def _mul32(a, b):
    return (np.asarray(a, dtype=np.uint64) * np.uint64(b)) & _M32


def _frame_key(seed, frame):
    return int(_mix((int(seed) * 0x9E3779B9 + int(frame) * 0x85EBCA6B + 0x71170000) & 0xFFFFFFFF))


def _pixel_hash(key, x, y, k):
    inner = _mix(np.uint64(key) ^ _mul32(y, 0xC2B2AE35))
    return _mix((inner + _mul32(x, 0x27D4EB2F) + np.uint64((k * 0x165667B1) & 0xFFFFFFFF)) & _M32)


def hash_frame(kind: str, w: int, h: int, seed: int = 0, frame: int = 0) -> np.ndarray:
    """Frame `frame` of the benchmark's synthetic stream, as timg_hip_synth_frames
    writes it into device memory."""
    k = HASH_KINDS[kind]
    key = _frame_key(seed, frame)
    x = np.arange(w, dtype=np.uint64)[None, :]
    y = np.arange(h, dtype=np.uint64)[:, None]
    h0 = _pixel_hash(key, x, y, 0)
    out = np.empty((h, w, 4), np.uint8)
    if k == 0:
        for c in range(4):
            out[..., c] = ((h0 >> np.uint64(8 * c)) & np.uint64(0xFF)).astype(np.uint8)
        return out

    def tri(ph):
        t = ph & np.uint64(0xFFFF)
        return np.where(t < 32768, t, np.uint64(65535) - t)

    for c in range(3):
        s = np.zeros((h, w), np.uint64)
        for t in range(3):
            i = (c * 3 + t) * 3
            a, b, d = (int(_mix((key + 0x1000 + j + i) & 0xFFFFFFFF)) for j in range(3))
            fx = ((128 + a % 1409) * 65536 // 256) // max(w, 1)
            fy = ((128 + b % 1409) * 65536 // 256) // max(h, 1)
            ph = d & 0xFFFF
            s += tri((_mul32(x, fx) + _mul32(y, fy) + np.uint64(ph)) & _M32)
        v = (s * np.uint64(255) // np.uint64(98301)).astype(np.int64)
        v += ((h0 >> np.uint64(8 * c)) & np.uint64(15)).astype(np.int64)
        v += ((h0 >> np.uint64(8 * c + 4)) & np.uint64(15)).astype(np.int64) - 15
        out[..., c] = np.clip(v, 0, 255).astype(np.uint8)
    if k == 1:
        out[..., 3] = 255
        return out
    xi = np.arange(w, dtype=np.int64)[None, :]
    yi = np.arange(h, dtype=np.int64)[:, None]
    dx, dy = 2 * xi - w + 1, 2 * yi - h + 1
    # (exact integers: Python ints via object arrays would be slow; uint64 holds every product here)
    d2 = (dx * dx).astype(np.uint64) * np.uint64(h * h) + (dy * dy).astype(np.uint64) * np.uint64(w * w)
    d2_max = max(1, w * w * h * h * 36 // 25)
    a = np.where(d2 >= np.uint64(d2_max), 0, 255 - (d2 * np.uint64(255) // np.uint64(d2_max)).astype(np.int64))
    m = min(w, h) // 8
    border = 64 if 64 < m else (m if m > 1 else 1)
    a[:border, :] = 0
    a[h - border:, :] = 0
    a[:, :border] = 0
    a[:, w - border:] = 0
    h1 = _pixel_hash(key, x, y, 1)
    special = np.array([0, 0x5F, 0x60, 0xFF], np.int64)[((h1 >> np.uint64(8)) & np.uint64(3)).astype(np.int64)]
    a = np.where(h1 % np.uint64(10) == 0, special, a)
    out[..., 3] = a.astype(np.uint8)
    return out
