"""CPU: the fixed-point accumulator arithmetic of csrc/kd6d_det.h (what makes the library's cross-workgroup sums
bitwise reproducible), built for the host with g++ from the SAME header the kernels include (tests/det_host.cpp)."""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def det(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("det") / "libdet.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "det_host.cpp")])
    L = ctypes.CDLL(so)
    for n in ("det_value_act", "det_value_grad", "det_sum_act", "det_sum_grad"):
        getattr(L, n).restype = ctypes.c_float
    L.det_value_act.argtypes = L.det_value_grad.argtypes = [ctypes.c_longlong] * 2
    return L


def _split(L, v, kind):
    out = (ctypes.c_longlong * 2)()
    getattr(L, "det_split_" + kind)(ctypes.c_float(v), out)
    return out[0], out[1]


def _sum(L, x, kind):
    x = np.ascontiguousarray(x, dtype=np.float32)
    return getattr(L, "det_sum_" + kind)(x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.c_longlong(len(x)))


@pytest.mark.parametrize("kind,E", [("act", 32), ("grad", 52)])
def test_image_of_an_addend_round_trips_within_half_a_unit(det, kind, E):
    rng = np.random.default_rng(E)
    vals = rng.standard_normal(4000).astype(np.float32) * np.float32(10.0) ** rng.integers(-14, 10, 4000).astype(np.float32)
    vals = np.concatenate([vals, np.float32([0.0, -0.0, 1.0, -1.0, 2.0 ** 15, 2.0 ** 15 - 2.0 ** -9, -2.0 ** 20, 1e-30,
                                             2.0 ** -(E + 1), 2.0 ** -E, -2.0 ** -(E + 2), 2.0 ** (47 - E), 2.0 ** (46 - E)])])
    for v in vals:
        lo, hi = _split(det, float(v), kind)
        assert abs(lo) < 2 ** 47 and abs(hi) < 2 ** 62
        assert lo == 0 or hi == 0 or (lo > 0) == (hi > 0)            # both words carry the addend's sign
        back = getattr(det, "det_value_" + kind)(lo, hi)
        assert abs(float(back) - float(v)) <= 0.5 * 2.0 ** -E + abs(float(v)) * 2.0 ** -24, (v, lo, hi, back)
        if abs(v) < 2.0 ** (47 - E):
            assert hi == 0                                           # one atomic in the usual case


@pytest.mark.parametrize("kind", ["act", "grad"])
def test_non_finite_and_absurd_addends_poison_the_accumulator(det, kind):
    for v in (float("inf"), float("-inf"), float("nan"), 3e38):
        lo, hi = _split(det, v, kind)
        assert math.isnan(getattr(det, "det_value_" + kind)(lo, hi))
    # and stay visible behind 2^15 more ordinary addends
    lo, hi = _split(det, float("nan"), kind)
    assert math.isnan(getattr(det, "det_value_" + kind)(lo + (1 << 46), hi + 12345))


def test_sum_does_not_depend_on_the_order_and_is_closer_than_fp32_accumulation(det):
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(200000) * 3).astype(np.float32)
    ref = float(np.sum(x.astype(np.float64)))
    a = _sum(det, x, "act")
    for seed in range(3):
        assert _sum(det, x[np.random.default_rng(seed).permutation(len(x))], "act") == a      # BITWISE
    assert abs(a - ref) <= abs(ref) * 2.0 ** -23 + 200000 * 2.0 ** -33
    serial = np.float32(0)
    for v in x[:20000]:
        serial = np.float32(serial + v)                   # what a chain of fp32 atomics computes, in one order
    assert abs(_sum(det, x[:20000], "act") - float(np.sum(x[:20000].astype(np.float64)))) <= \
        abs(float(serial) - float(np.sum(x[:20000].astype(np.float64)))) + 1e-6
    g = (rng.standard_normal(100000) * 1e-7).astype(np.float32)            # gradient-sized addends
    gs = _sum(det, g, "grad")
    assert gs == _sum(det, g[::-1], "grad")
    assert abs(gs - float(np.sum(g.astype(np.float64)))) <= 1e-12
    big = np.float32([3e4, -2.9e4, 1e5, 7.0, 2.0 ** 20] * 1000)           # spill into the hi word, mixed with small ones
    assert _sum(det, big, "act") == _sum(det, big[::-1], "act")
    assert abs(_sum(det, big, "act") - float(np.sum(big.astype(np.float64)))) <= float(np.sum(big.astype(np.float64))) * 2.0 ** -23
