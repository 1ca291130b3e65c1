"""CPU-only: the product's host-side plan builder (timg_amd/csrc/
resample_plan.cc, exported through a debug symbol) must produce tap tables
bit-identical to the oracle's restatement of stb_image_resize2."""
import numpy as np
import pytest

import oracle_lib

CASES = [(3840, 2160, 800, 450), (3840, 2160, 200, 56), (7680, 4320, 800, 450),
         (640, 480, 67, 50), (64, 64, 64, 64), (100, 100, 10, 10), (50, 40, 120, 90),
         (1000, 1000, 100, 100), (1, 1, 1, 1), (1, 7, 9, 1), (1920, 1080, 67, 50)]


def _same(a, b):
    return a["header"] == b["header"] and all(
        np.array_equal(a[k], b[k]) for k in ("h_taps", "h_coeff", "v_cnt", "v_rows", "v_coeff"))


@pytest.mark.parametrize("sw,sh,dw,dh", CASES)
@pytest.mark.parametrize("filt", [0, 2])
def test_named_geometries(oracle, sw, sh, dw, dh, filt):
    assert _same(oracle.plan_dump(sw, sh, dw, dh, filt), oracle_lib.product_plan_dump(sw, sh, dw, dh, filt))


def test_random_geometries(oracle):
    rng = np.random.default_rng(2)
    for _ in range(1500):
        sw, sh, dw, dh = (int(v) for v in rng.integers(1, 400, 4))
        filt = int(rng.choice([0, 2]))
        assert _same(oracle.plan_dump(sw, sh, dw, dh, filt),
                     oracle_lib.product_plan_dump(sw, sh, dw, dh, filt)), (sw, sh, dw, dh, filt)


def test_baseline_plan_shapes(oracle):
    """The pass order / algorithm stb picks at the BASELINE configs."""
    i = oracle.plan_info(3840, 2160, 800, 450)
    assert (i["vertical_first"], i["h_widest"], i["v_is_gather"], i["v_widest"]) == (1, 20, 2, 20)
    i = oracle.plan_info(3840, 2160, 200, 56)
    assert (i["vertical_first"], i["v_is_gather"]) == (1, 0) and i["v_widest"] == 155
    i = oracle.plan_info(7680, 4320, 800, 450)
    assert (i["vertical_first"], i["v_is_gather"], i["h_widest"]) == (0, 0, 39)
