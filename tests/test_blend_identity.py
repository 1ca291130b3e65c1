"""The alpha-compose epilogue (timg_amd/csrc/pixel_math.h, BlendChannelByte) replaces the reference's
`sqrtf((c^2 a + bg^2 (255 - a)) / 255)` + clamp + truncation (src/framebuffer.h:150-161) by an exact integer
identity.  Enumerated here over every value the numerator can take (integers 0 .. 255 * 255^2, all exact in
fp32): the fp32 pipeline with correctly rounded divide and square root -- numpy's float32 ops are -- yields
floor(sqrt(x / 255)), and the device's recipe (approximate root, truncate, two comparisons against 255 k^2 and
255 (k + 1)^2) yields the same byte however the approximate root errs within a few ulps."""
import numpy as np

X_MAX = 255 * 255 * 255


def _reference_bytes():
    x = np.arange(0, X_MAX + 1, dtype=np.float32)
    s = np.sqrt(x / np.float32(255.0))
    return np.where(s > np.float32(255.0), 255, s.astype(np.uint32)).astype(np.int64)


def test_reference_pipeline_is_floor_sqrt_of_the_quotient():
    f = _reference_bytes()
    assert np.all(np.diff(f) >= 0)
    k = np.arange(1, 256, dtype=np.int64)
    first = np.searchsorted(f, k, side="left")  # smallest x whose byte reaches k
    assert np.array_equal(first, 255 * k * k)
    assert f[0] == 0 and f[-1] == 255


def _device_recipe(x, root):
    k = np.trunc(root).astype(np.float32)
    k = np.where((k * k) * np.float32(255.0) > x, k - np.float32(1.0), k).astype(np.float32)
    k1 = k + np.float32(1.0)
    k = np.where((k1 * k1) * np.float32(255.0) <= x, k1, k)
    return k.astype(np.int64)


def test_device_recipe_survives_an_inexact_root():
    f = _reference_bytes()
    x = np.arange(0, X_MAX + 1, dtype=np.float32)
    y = x * np.float32(1.0 / 255.0)
    exact = np.sqrt(y.astype(np.float64))
    for rel in (0.0, -3e-7, +3e-7, -1e-6, +1e-6):  # a hardware root is good to ~1 ulp (6e-8); these are 5-17 ulps
        root = (exact * (1.0 + rel)).astype(np.float32)
        root = np.maximum(root, np.float32(0.0))
        assert np.array_equal(_device_recipe(x, root), f), rel


def test_checkerboard_column_division_by_multiplication():
    """CheckerAlt (timg_amd/csrc/device_plan.h): x / pw as mulhi(x, ceil(2^32 / pw)) for pixel columns and cell
    widths below 65 536 -- exact for every column and a sweep of cell widths incl. the extremes."""
    x = np.arange(65536, dtype=np.uint64)
    for pw in list(range(2, 300)) + [511, 512, 513, 1000, 4095, 4096, 4097, 32767, 32768, 65535]:
        magic = (2 ** 32 + pw - 1) // pw
        assert magic < 2 ** 32
        assert np.array_equal((x * np.uint64(magic)) >> np.uint64(32), x // np.uint64(pw)), pw
