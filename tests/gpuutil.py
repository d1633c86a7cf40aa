"""Helpers shared by the GPU parity tests."""
import numpy as np

REL = 1e-6          # BASELINE.json north_star: redshift / flux / polarization angle within 1e-6 relative


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
