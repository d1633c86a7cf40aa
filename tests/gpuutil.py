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


def assert_flux(flux, r, ref_flux, a, what="", rtol=REL, M=10.0, mdot=0.1, alpha_visc=0.1):
    """Flux parity WITHOUT a floor (round 6; until round 5 every flux assertion carried max(F, 1e-9 F_peak) in its denominator).

    The Novikov-Thorne closed form (ref src/sim5disk-nt.c:129-135) cancels towards the inner edge: within ~1e-5 r_g of it the
    REFERENCE's own value moves by more than 1e-6 when its argument moves by ONE ulp (profiles/r06_flux_edge_noise.json), so a
    pixel whose radius agrees with the reference's to 1e-13 -- hundreds of ulps -- cannot be held to the reference's pixel
    value there by anybody; what can be held, to the bar and at every lit pixel, is the FUNCTION on the same input bits:
      (1) flux == disk_nt_flux of the reference evaluated AT THE GPU's OWN RADII (the live oracle/_ref library on the box, our
          byte-pinned restatement when it is absent), within rtol relative, no floor, every element;
      (2) against the reference's pixel values `ref_flux` (its own radii) the difference stays within rtol * F plus what the
          reference itself moves between the two radii, |F_ref(r_gpu) - F_ref(r_ref)| -- for all but a handful of pixels per
          image that term is far below 1e-6 F.
    The radii themselves are asserted by the caller (1e-6 by the bar; 1e-12 measured).  `flux`, `r`, `ref_flux`: same shape;
    elements whose r is NaN (no hit) must have no flux on either side."""
    import oraclelib as ol
    flux, r, ref_flux = (np.asarray(v, np.float64) for v in (flux, r, ref_flux))
    assert flux.shape == r.shape == ref_flux.shape, (flux.shape, r.shape, ref_flux.shape)
    lit = np.isfinite(r)
    assert not np.any(np.nan_to_num(flux[~lit]) != 0) and not np.any(np.nan_to_num(ref_flux[~lit]) != 0), "%s: flux where there is no hit" % what
    if not lit.any():
        return 0.0
    own = ol.cpu_disk_flux(r[lit], a, M=M, mdot=mdot, alpha_visc=alpha_visc)
    e = rel_err(flux[lit], own)
    assert e <= rtol, "%s: flux against the reference's disk_nt_flux at the same radii: max rel err %.3e > %.1e (no floor)" % (what, e, rtol)
    d = np.abs(flux[lit] - ref_flux[lit])
    allow = 1.000001 * (rtol * np.abs(own) + np.abs(own - ref_flux[lit]))
    bad = d > allow
    assert not bad.any(), "%s: %d fluxes off the reference's pixel values by more than the bar plus the reference's own movement" % (what, int(bad.sum()))
    return e
