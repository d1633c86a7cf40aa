"""Batched counterpart of SIM5's Python ray tracer for disk photospheres
(ref: python/sim5diskraytrace.py: DiskRaytrace.geodesic :214-253, __find_surface :257-335, .image :138-210,
__tetrad :340-348, __gfactor :353-361, __emission_angle :377-390; python/sim5diskmodel.py:
DiskModel_ThinDisk :70-96).

Same class and method names, same returned quantities, but every SIM5 call is made ONCE for all rays of
the image through the batch entry points of the C-ABI (sim5_amd/capi.py) instead of once per pixel
through SWIG; the surface search for geometrically thick disks is one kernel (sim5gpu_disk_surface_rays).
image() covers flat disks and disks with a tabulated photosphere H(R).
"""
import math

import numpy as np

from . import capi as _c
from .sim5lib import grav_radius, parsec


class DiskModel_ThinDisk:
    """Novikov-Thorne disk (ref: python/sim5diskmodel.py:70-96); mdot-parametrised set-up only."""

    def __init__(self, bh_mass, bh_spin, mdot, alpha, options=0):
        _c.disk_nt_setup(bh_mass, bh_spin, mdot, alpha, options or 0)
        self.name = "Novikov-Thorne"
        self.mdot = mdot
        self.bh_spin = bh_spin
        self.r_min = _c.disk_nt_r_min()

    def flux(self, R):
        return _c.disk_nt_flux(np.atleast_1d(R))

    def t_eff(self, R):
        return (self.flux(R) / 5.670400e-05) ** 0.25          # ref: sim5diskmodel.py:47

    def l(self, R):
        return _c.disk_nt_ell(np.atleast_1d(R))

    def vr(self, R):
        return np.zeros(np.shape(np.atleast_1d(R)))

    def h(self, R):
        return np.zeros(np.shape(np.atleast_1d(R)))

    def dhdr(self, R):
        return np.zeros(np.shape(np.atleast_1d(R)))


class DiskModel_Surface(DiskModel_ThinDisk):
    """A disk with a geometrically thick photosphere given as a table H(R) (R ascending): linear interpolation,
    H[0] below the table, constant opening angle beyond it -- the definition the surface-search kernel uses.
    Flux and angular momentum are the Novikov-Thorne ones; `vr` is an optional callable of R.  A model with
    other profiles derives from this class and overrides flux / l / vr (they receive arrays)."""

    def __init__(self, bh_mass, bh_spin, mdot, alpha, table_R, table_H, vr=None, table_vr=None):
        super().__init__(bh_mass, bh_spin, mdot, alpha)
        self.name = "tabulated surface"
        self.bh_mass = bh_mass
        self.tR = np.ascontiguousarray(table_R, dtype=np.float64)
        self.tH = np.ascontiguousarray(table_H, dtype=np.float64)
        self._vr = vr
        # radial velocity on the nodes of the table, interpolated like the surface; with it (or with no radial
        # velocity at all) DiskRaytrace.image() runs as ONE kernel (sim5gpu_disk_surface_frame)
        self.tV = None if table_vr is None else np.ascontiguousarray(table_vr, dtype=np.float64)
        self.fused = (vr is None) and type(self).flux is DiskModel_ThinDisk.flux and type(self).l is DiskModel_ThinDisk.l

    def surface_table(self):
        return self.tR, self.tH

    def h(self, R):
        R = np.atleast_1d(np.asarray(R, dtype=np.float64))
        inner = np.interp(R, self.tR, self.tH)                   # clamps to H[0] below the table
        return np.where(R >= self.tR[-1], self.tH[-1] * (R / self.tR[-1]), inner)

    def dhdr(self, R):
        R = np.atleast_1d(np.asarray(R, dtype=np.float64))
        hi = np.clip(np.searchsorted(self.tR, R, side="left"), 1, self.tR.size - 1)
        slope = (self.tH[hi] - self.tH[hi - 1]) / (self.tR[hi] - self.tR[hi - 1])
        slope = np.where(R > self.tR[0], slope, 0.0)
        return np.where(R >= self.tR[-1], self.tH[-1] / self.tR[-1], slope)

    def vr(self, R):
        R = np.atleast_1d(np.asarray(R, dtype=np.float64))
        if self._vr is not None:
            return np.asarray(self._vr(R), dtype=np.float64)
        if self.tV is not None:
            return np.interp(R, self.tR, self.tV)                 # clamps to the end nodes outside the table
        return np.zeros(R.shape)


class DiskRaytrace:
    def __init__(self, bh_mass, bh_spin, bh_dist, disk_model, spectral_model=None):
        if bh_spin < 1e-4:
            bh_spin = 1e-4                                        # ref :32
        self.bh_mass, self.bh_spin, self.bh_dist = bh_mass, bh_spin, bh_dist
        self.disk, self.spectra = disk_model, spectral_model

    def geodesic(self, incl, alpha, beta, flat=True):
        """Arrays alpha, beta -> dict(ok, r, m, P, k[n,4], gd records).  incl in radians (ref :214-253)."""
        alpha = np.ascontiguousarray(alpha, dtype=np.float64).ravel()
        beta = np.ascontiguousarray(beta, dtype=np.float64).ravel()
        if not flat:
            # thick disk: the reference's __find_surface (:257-335) as one kernel; the disk model supplies
            # its photosphere as a table through surface_table() -> (R[], H[])
            tR, tH = self.disk.surface_table()
            s = _c.disk_surface_rays(self.bh_spin, incl, tR, tH, alpha, beta, checked=None)      # the host copy is checked here: no read-back
            good = s["status"] == 1
            return {"ok": good, "r": np.where(good, s["r"], 0.0), "m": np.where(good, s["m"], 0.0),
                    "P": s["P"], "k": s["k"], "gd": None}
        gd, err, ok = _c.geodesic_init_inf(incl, self.bh_spin, alpha, beta)
        good = err == 0
        P = np.full(alpha.size, np.nan); r = np.full(alpha.size, np.nan)
        if good.any():
            P[good] = _c.geodesic_find_midplane_crossing(gd[good], 0)
            r[good] = _c.geodesic_position_rad(gd[good], P[good])
        good &= ~np.isnan(r)                                      # ref :248
        k = np.full((alpha.size, 4), np.nan)
        if good.any():
            g = gd[good]
            k[good] = _c.photon_momentum(self.bh_spin, r[good], 0.0, g["l"], g["q"], g["Rpc"] - P[good], 1.0)   # ref :250
        return {"ok": good, "r": np.where(good, r, 0.0), "m": np.zeros(alpha.size), "P": P, "k": k, "gd": gd}

    def _tetrad(self, r, m):                                      # ref :340-348
        R = r * np.sqrt(1. - m * m)
        metric = _c.kerr_metric(self.bh_spin, r, m)
        Om = _c.Omega_from_ell(self.disk.l(R), metric)
        dhdr = np.where(m > 0.0, self.disk.dhdr(R), 0.0)
        return _c.tetrad_surface(metric, Om, self.disk.vr(R), dhdr), metric

    def _gfactor(self, k, tetrad, metric):                        # ref :353-361
        n = k.shape[0]
        U = _c.on2bl(np.tile([1.0, 0.0, 0.0, 0.0], (n, 1)), tetrad)
        g = (k[:, 0] * metric["g00"] + k[:, 3] * metric["g03"]) / _c.dotprod(k, U, metric)
        return np.where(g > 0.0, g, 0.0)

    def _emission_angle(self, k, tetrad, metric):                 # ref :377-390
        n = k.shape[0]
        U = _c.on2bl(np.tile([1.0, 0.0, 0.0, 0.0], (n, 1)), tetrad)
        N = _c.on2bl(np.tile([0.0, 0.0, 1.0, 0.0], (n, 1)), tetrad)
        mue = _c.dotprod(k, N, metric) / _c.dotprod(k, U, metric)
        return np.where((mue < 0.0) & (mue > -1e-2), 1e-3, mue)

    def spectrum_image(self, incl, rmax, N, energies, limbdk=1, hardening=1.7):
        """Observed spectrum [erg/s/cm2/keV] of the N x N pixel grid: the accumulation of the reference's
        spectrum() (ref :96-123: black body at E/g, times g^3 dOmega) fused with the ray tracing in one kernel."""
        incl = math.radians(max(1.0, incl))
        d = _c.image_desc(N, N, self.bh_spin, incl, rmax=rmax, bh_mass=self.bh_mass, mdot=self.disk.mdot,
                          disk_spin=getattr(self.disk, "bh_spin", -1.0))
        dOmega = (2.0 * rmax / N) ** 2 * ((self.bh_mass * grav_radius) / (self.bh_dist * parsec * 1e3)) ** 2
        return _c.disk_spectrum(d, energies, hardening=hardening, limb_darkening=limbdk) * dOmega

    def image(self, incl, rmax, N, limbdk=1, fused=True):
        """Disk image (ref :138-210).  incl in degrees; returns the reference's dict of N x N arrays
        (NaN where the reference leaves None).  fused=False forces the call-by-call path for a DiskModel_Surface."""
        incl = math.radians(max(1.0, incl))
        c = ((np.arange(N) + .5) / N - 0.5) * 2.0 * rmax
        alpha = np.tile(c, N); beta = np.repeat(c, N)
        dOmega = (2.0 * rmax / N) ** 2 / ((self.bh_mass * grav_radius) / (self.bh_dist * parsec * 1e3)) ** 2
        flat = bool(np.all(np.asarray(self.disk.h(1e5)) == 0.0))         # ref :176
        out = {k: np.full(N * N, np.nan) for k in ("flux", "gfactor", "mue", "T", "R", "H", "V")}
        if not flat and getattr(self.disk, "fused", False) and fused:
            # surface search and local frame in one kernel; the selection rules of ref :174-198 on the host
            s = _c.disk_surface_frame(self.bh_spin, incl, self.disk.bh_mass, self.disk.mdot, self.disk.tR, self.disk.tH,
                                      alpha, beta, table_vr=self.disk.tV, disk_spin=self.disk.bh_spin, checked=None)
            r, m = s["r"], s["m"]
            R = r * np.sqrt(1. - m * m)
            sel = (s["status"] == 1) & (s["flux"] != 0.0) & (s["g"] > 0.0)
            e = s["mue"][sel]
            l = (0.5 + 0.75 * e) if limbdk > 0 else np.ones_like(e)
            e = np.where(e < 0.0, 0.0001, np.where(e > 1.0, 0.9999, e))
            g = s["g"][sel]; F = s["flux"][sel]
            out["flux"][sel] = F * g ** 4 * l * dOmega
            out["gfactor"][sel] = g
            out["mue"][sel] = np.degrees(np.arccos(e))
            out["T"][sel] = (F / 5.670400e-05) ** 0.25
            out["R"][sel] = R[sel]
            out["H"][sel] = (r * m)[sel]
            out["V"][sel] = np.asarray(self.disk.vr(R[sel]))
            return {k: v.reshape(N, N) for k, v in out.items()}
        geo = self.geodesic(incl, alpha, beta, flat=flat)
        sel = geo["ok"].copy()
        r, m = geo["r"], geo["m"]
        R = r * np.sqrt(1. - m * m)
        F = np.zeros(N * N)
        F[sel] = self.disk.flux(R[sel])
        sel &= F != 0.0                                           # ref :181
        if sel.any():
            tet, met = self._tetrad(r[sel], m[sel])
            k = geo["k"][sel]
            g = self._gfactor(k, tet, met)
            e = self._emission_angle(k, tet, met)
            l = (0.5 + 0.75 * e) if limbdk > 0 else np.ones_like(e)      # before the clamp, as ref :186-192
            e = np.where(e < 0.0, 0.0001, np.where(e > 1.0, 0.9999, e))
            keep = g > 0.0
            idx = np.nonzero(sel)[0][keep]
            out["flux"][idx] = (F[sel] * g ** 4 * l * dOmega)[keep]
            out["gfactor"][idx] = g[keep]
            out["mue"][idx] = np.degrees(np.arccos(e[keep]))
            out["T"][idx] = np.asarray(self.disk.t_eff(R[sel]))[keep]
            out["R"][idx] = R[sel][keep]
            out["H"][idx] = (r * m)[sel][keep]
            out["V"][idx] = np.asarray(self.disk.vr(R[sel]))[keep]
        return {k: v.reshape(N, N) for k, v in out.items()}
