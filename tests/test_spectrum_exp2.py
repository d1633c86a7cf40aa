"""The Planck factor's 2^f polynomial of the fast spectrum kernel (sim5_amd/csrc/k_spectrum.hip planck_sum): the literals in the
kernel source are the coefficients tests/tools/exp2_coefficients.py derives, and their relative error on 2^f - 1 is what the
comment says -- a CPU check of the numbers the GPU test then holds to the strict kernel at 1e-6."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _literals():
    src = open(os.path.join(ROOT, "sim5_amd", "csrc", "k_spectrum.hip")).read()
    body = src[src.index("S5_DEV double planck_sum("):]
    m = re.search(r"constexpr double C1 = ([0-9.e+-]+), C2 = ([0-9.e+-]+), C3 = ([0-9.e+-]+), C4 = ([0-9.e+-]+),\s*C5 = ([0-9.e+-]+), C6 = ([0-9.e+-]+);", body)
    assert m, "coefficient line of planck_sum not found"
    return [float(x) for x in m.groups()]


def test_kernel_literals_are_the_derived_coefficients():
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import exp2_coefficients as ec
    co, worst = ec.remez(ec.DEG)
    assert float(worst) < 1.1e-8
    lit = _literals()
    assert len(lit) == len(co)
    for a, b in zip(lit, co):
        assert abs(a - float(b)) <= 2e-16 * abs(a), (a, float(b))


def test_relative_error_of_two_to_the_f_minus_one():
    c = _literals()
    f = np.linspace(-0.5, 0.5, 200001)
    f = f[f != 0]
    g = np.zeros_like(f)
    for ck in c[::-1]:
        g = g * f + ck
    err = np.abs(f * g / np.expm1(f * np.log(2.0)) - 1)
    assert err.max() < 1.1e-8, err.max()


def test_integer_part_from_the_low_word():
    """n = low 32 bits of the double t + 1.5 2^52, as a two's complement integer, for |t| < 2^31: the identity the kernel's
    v_ldexp relies on (round to nearest even included)."""
    rng = np.random.default_rng(5)
    t = np.concatenate([rng.uniform(-2.0**31 + 1, 2.0**31 - 1, 100000), rng.uniform(-40, 40, 100000),
                        np.array([0.5, 1.5, 2.5, -0.5, -1.5, 1073741824.0, -1073741824.0, 0.0])])
    tm = t + 6755399441055744.0
    low = (tm.view(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)
    assert np.array_equal(low.astype(np.float64), np.rint(t))
    assert np.array_equal(tm - 6755399441055744.0, np.rint(t))
