"""Helpers shared by the GPU parity tests."""
import numpy as np

import math

REL = 1e-6          # BASELINE.json north_star: redshift / flux / polarization angle within 1e-6 relative


def deg2rad(deg):
    """degrees -> radians as the reference's macro forms them (ref src/sim5math.h:50: (a)/180.0*M_PI), which is what the CPU
    checker's drivers do with an inclination in degrees.  math.radians(x) = x * (pi/180) differs from it in the last bit for most
    x -- and the last bit of the inclination reaches cos i, q and, on the central column of an odd-width image, the class of a
    pixel: a GPU job and the CPU run it is compared with must be given the SAME radians."""
    return deg / 180.0 * math.pi


def rel_err(a, b, floor=0.0):
    """max over elements of |a-b| / max(|b|, floor); NaNs must coincide."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), "NaN pattern differs in %d places" % (na != nb).sum()
    m = ~na
    if not m.any():
        return 0.0
    d = np.abs(a[m] - b[m])
    den = np.maximum(np.abs(b[m]), floor)
    with np.errstate(divide="ignore", invalid="ignore"):
        e = np.where(d == 0, 0.0, d / den)
    return float(np.max(e))


def assert_close(a, b, rtol=REL, floor=0.0, what=""):
    e = rel_err(a, b, floor)
    assert e <= rtol, "%s: max rel err %.3e > %.1e" % (what, e, rtol)
    return e
