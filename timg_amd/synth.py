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
