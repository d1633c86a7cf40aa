"""sim5_amd/diskraytrace.py (the batched Python counterpart of the reference's DiskRaytrace) against the
golden vectors captured from the reference's own Python class (oracle/gen_golden_py.py).

CPU part: the host-side logic with the batch calls served by the CPU oracle (tests/oracle_capi.py).
GPU part: the same class on the real library."""
import math

import numpy as np
import pytest


def check_against_golden(drt_module, g, rtol):
    cases = g["cases"]
    for ci, (a, inc) in enumerate(cases):
        disk = drt_module.DiskModel_ThinDisk(10.0, float(a), 0.1, 0.1)
        rt = drt_module.DiskRaytrace(10.0, float(a), 10.0, disk, None)
        rmax = float(g["rmax%d" % ci][0])
        N = g["img%d_flux" % ci].shape[0]
        img = rt.image(float(inc), rmax, N)
        for k in ("flux", "gfactor", "mue", "T", "R", "H"):
            ref = g["img%d_%s" % (ci, k)]
            got = img[k]
            assert np.array_equal(np.isnan(ref), np.isnan(got)), (ci, k, "pixels set differ")
            m = ~np.isnan(ref)
            if not m.any():
                continue
            top = np.abs(ref[m]).max()
            scale = np.maximum(np.abs(ref[m]), 1e-9 * top) if top > 0 else np.ones(m.sum())
            err = np.max(np.abs(got[m] - ref[m]) / scale)
            assert err < rtol, (ci, k, err)
        c = ((np.arange(N) + .5) / N - 0.5) * 2.0 * rmax
        geo = rt.geodesic(math.radians(float(inc)), np.tile(c, N), np.repeat(c, N), flat=True)
        rr = g["geo%d_r" % ci].ravel(); kk = g["geo%d_k" % ci].reshape(-1, 4)
        ok = ~np.isnan(rr)
        assert np.array_equal(ok, geo["ok"])
        assert np.max(np.abs(geo["r"][ok] / rr[ok] - 1)) < rtol
        assert np.max(np.abs(geo["k"][ok] - kk[ok]) / np.maximum(np.abs(kk[ok]), 1e-6)) < max(rtol, 1e-9)


def test_host_logic_with_oracle_backend(golden, monkeypatch):
    import oracle_capi
    import sim5_amd.diskraytrace as drt
    monkeypatch.setattr(drt, "_c", oracle_capi)
    check_against_golden(drt, golden("py_diskraytrace.npz"), rtol=1e-12)


def test_host_logic_with_oracle_backend_over_spins_sizes_and_fields(golden, monkeypatch):
    """The same over 30 more jobs of the reference's class (oracle/gen_golden_thin.py: spins 0 .. 0.998, inclinations 5 .. 86
    degrees, image sizes 10 .. 22 -- odd ones included --, fields of view r_ms + 8 / 20 / 50)."""
    import oracle_capi
    import sim5_amd.diskraytrace as drt
    monkeypatch.setattr(drt, "_c", oracle_capi)
    check_against_golden(drt, golden("py_thin_more.npz"), rtol=1e-12)


def test_sim5lib_module_has_the_swig_names():
    import sim5_amd.sim5lib as s
    need = """r_ms r_bh disk_nt_setup disk_nt_r_min disk_nt_flux disk_nt_ell geodesic intp doublep doubleArray
              sim5metric sim5tetrad sim5vector double_array_getitem geodesic_init_inf geodesic_find_midplane_crossing
              geodesic_position_rad geodesic_position_pol geodesic_P_int geodesic_follow photon_momentum kerr_metric
              tetrad_surface Omega_from_ell on2bl dotprod grav_radius parsec solar_mass grav_const""".split()
    missing = [n for n in need if not hasattr(s, n)]          # names used by ref python/sim5diskraytrace.py
    assert not missing, missing


@pytest.mark.gpu
def test_batched_diskraytrace_on_gpu(golden, capi):
    import sim5_amd.diskraytrace as drt
    check_against_golden(drt, golden("py_diskraytrace.npz"), rtol=1e-6)
    check_against_golden(drt, golden("py_thin_more.npz"), rtol=1e-6)          # 30 more jobs: oracle/gen_golden_thin.py


@pytest.mark.gpu
def test_scalar_sim5lib_module_on_gpu(golden, capi):
    """The reference's per-pixel call sequence (python/sim5diskraytrace.py:228-250, 340-361) through the
    SWIG-name module."""
    import sim5_amd.sim5lib as s
    g = golden("py_diskraytrace.npz")
    ci = 3
    a, inc = g["cases"][ci]
    rmax = float(g["rmax%d" % ci][0]); N = 16
    s.disk_nt_setup(10.0, float(a), 0.1, 0.1, 0)
    for (y, x) in [(2, 3), (8, 8), (12, 5), (15, 15)]:
        al = ((x + .5) / N - 0.5) * 2.0 * rmax; be = ((y + .5) / N - 0.5) * 2.0 * rmax
        status = s.intp(); gd = s.geodesic(); k = s.doubleArray(4)
        s.geodesic_init_inf(math.radians(float(inc)), float(a), al, be, gd, status)
        ref_r = g["geo%d_r" % ci][y, x]
        if status.value() != 0:
            assert np.isnan(ref_r); continue
        P = s.geodesic_find_midplane_crossing(gd, 0)
        r = s.geodesic_position_rad(gd, P)
        if math.isnan(r):
            assert np.isnan(ref_r); continue
        assert abs(r / ref_r - 1) < 1e-9
        s.photon_momentum(float(a), r, 0.0, gd.l, gd.q, gd.Rpc - P, 1.0, k)
        assert np.allclose([k[i] for i in range(4)], g["geo%d_k" % ci][y, x], rtol=1e-8, atol=1e-12)
        metric = s.sim5metric(); tetrad = s.sim5tetrad()
        s.kerr_metric(float(a), r, 0.0, metric)
        s.tetrad_surface(metric, s.Omega_from_ell(s.disk_nt_ell(r), metric), 0.0, 0.0, tetrad)
        U = s.doubleArray(4)
        s.on2bl(s.sim5vector((1, 0, 0, 0)), U, tetrad)
        m = tetrad.metric
        gf = (s.double_array_getitem(k, 0) * m.g00 + s.double_array_getitem(k, 3) * m.g03) / s.dotprod(k, U, m)
        ref_g = g["img%d_gfactor" % ci][y, x]
        if not np.isnan(ref_g):
            assert abs(gf / ref_g - 1) < 1e-8


@pytest.mark.gpu
def test_fused_spectrum_kernel(golden, capi):
    """sim5gpu_disk_spectrum against the reference's Python classes: DiskRaytrace.image quantities fed to
    DiskSpectrum_BlackBody.spectrum and accumulated as DiskRaytrace.spectrum does (oracle/gen_golden_py.py)."""
    g = golden("py_diskraytrace.npz")
    E = g["spec_E"]
    for ci, (a, inc) in enumerate(g["cases"]):
        rmax = float(g["rmax%d" % ci][0])
        for (tag, limb, hard) in (("a", 1, 1.7), ("b", 0, 1.0)):
            ref = g["spec%d%s" % (ci, tag)]
            for strict in (False, True):
                d = capi.image_desc(16, 16, max(float(a), 1e-4), math.radians(float(inc)), rmax=rmax,
                                    disk_spin=float(a), strict=strict)
                got = capi.disk_spectrum(d, E, hardening=hard, limb_darkening=limb)
                err = np.max(np.abs(got - ref) / np.maximum(ref, 1e-9 * ref.max()))
                assert err < 1e-6, (ci, tag, strict, err)
    # a size that is not a multiple of the tile, more energies than one pass holds, run twice: deterministic
    d = capi.image_desc(100, 60, 0.9, 1.2)
    E2 = 10.0 ** np.linspace(-2, 2, 300)
    s1 = capi.disk_spectrum(d, E2); s2 = capi.disk_spectrum(d, E2)
    assert np.array_equal(s1, s2) and np.isfinite(s1).all() and s1.max() > 0
    # consistency with the image: total of the spectrum bins equals the per-pixel sum done on the host
    img = capi.disk_image(capi.image_desc(100, 60, 0.9, 1.2, max_order=1, rms=1e-9), full=True)
    assert (img["flux"] > 0).sum() > 1000


@pytest.mark.gpu
def test_fused_spectrum_kernel_over_disks_and_grids(golden, capi):
    """sim5gpu_disk_spectrum against the reference's Python classes over what the six cases above hold fixed
    (oracle/gen_golden_spectrum.py, 36 jobs): image sizes 12 .. 28, spins 0 .. 0.998, inclinations 5 .. 86 degrees, masses 5 ..
    1e8 and accretion rates 0.01 .. 1 (spectra that peak from the optical to hard X-rays), fields of view r_ms + 8 / 20 / 60,
    energy grids in logarithmic and in EQUAL steps (24 .. 100 energies: the general loop and the recurrence), every
    combination of limb darkening and hardening.  Both variants, every bin within 1e-6 (bins below 1e-9 of the peak: of
    that)."""
    g = golden("py_spectrum_more.npz")
    worst = {False: 0.0, True: 0.0}
    for ci, (a, inc, N, mass, mdot, rmax, limb, hard, equal) in enumerate(g["cases"]):
        E, ref = g["c%d_E" % ci], g["c%d_spec" % ci]
        assert ref.max() > 0
        for strict in (False, True):
            d = capi.image_desc(int(N), int(N), max(float(a), 1e-4), math.radians(float(inc)), rmax=float(rmax), disk_spin=float(a),
                                bh_mass=float(mass), mdot=float(mdot), alpha_visc=0.1, strict=strict)
            got = capi.disk_spectrum(d, E, hardening=float(hard), limb_darkening=int(limb))
            err = float(np.max(np.abs(got - ref) / np.maximum(ref, 1e-9 * ref.max())))
            worst[strict] = max(worst[strict], err)
            assert err < 1e-6, (ci, "strict" if strict else "fast", err, [float(x) for x in g["cases"][ci]])
    print("spectrum against the reference's Python classes, %d jobs: worst bin fast %.1e strict %.1e" % (len(g["cases"]), worst[False], worst[True]))


@pytest.mark.gpu
@pytest.mark.parametrize("n_energies,lo,hi", [(128, -1.0, 1.5), (64, -2.0, 2.0), (256, -3.0, 3.0), (300, -2.0, 2.0), (17, -1.0, 1.0),
                                              (128, -3.0, 12.0), (40, -6.0, 13.0), (1, 0.0, 0.0)],
                         ids=["128", "64", "256", "300-two-passes", "17-run-time-stride", "128-to-1e12-keV", "40-to-1e13-keV", "one"])
def test_spectrum_fast_against_strict_over_energy_grids(capi, n_energies, lo, hi):
    """The fast kernel's Planck factor (k_spectrum.hip planck_sum: 2^n from the low word of t + 1.5 2^52, a degree-6 2^f, the
    reciprocal seed) against the strict kernel's full-precision evaluation (ref python/sim5diskspectrum.py:54-88), for every
    loop form: compile-time strides 1, 2, 4 (256, 128, 64 energies per pass), the run-time stride (< 64), several passes, and grids
    whose largest energy takes x log2(e) beyond the 32-bit range of that low word (the bounded form of the loop: exactly-zero
    terms, never a wrapped exponent).  1e-6 of the bin, bins below 1e-280 of the peak compared as zeros."""
    d_fast = capi.image_desc(160, 96, 0.9, 1.2)
    d_strict = capi.image_desc(160, 96, 0.9, 1.2, strict=True)
    E = 10.0 ** np.linspace(lo, hi, n_energies)
    f = capi.disk_spectrum(d_fast, E)
    s = capi.disk_spectrum(d_strict, E)
    assert np.isfinite(f).all() and np.isfinite(s).all() and s.max() > 0
    live = s > 1e-280 * s.max()
    assert np.array_equal(f[~live] > 1e-270 * s.max(), np.zeros((~live).sum(), bool)), "bins that must be (next to) nothing"
    err = np.max(np.abs(f[live] / s[live] - 1))
    assert err < 1e-6, (n_energies, lo, hi, err, int(np.argmax(np.abs(f[live] / s[live] - 1))))
    if hi > 9:
        assert (f[E > 1e6] == 0).all(), "x beyond the exponent range is an exact zero"
    # a row set that is not symmetric about the middle (the unpaired instantiation, 256 pixels per workgroup), and the two
    # parts of the image add up to the whole
    top = capi.disk_spectrum(capi.image_desc(160, 96, 0.9, 1.2, y0=0, y1=37), E)
    rest = capi.disk_spectrum(capi.image_desc(160, 96, 0.9, 1.2, y0=37, y1=96), E)
    top_s = capi.disk_spectrum(capi.image_desc(160, 96, 0.9, 1.2, y0=0, y1=37, strict=True), E)
    lt = top_s > 1e-280 * max(top_s.max(), 1e-300)
    assert np.max(np.abs(top[lt] / top_s[lt] - 1)) < 1e-6
    # (the pixels meet other pixels in the groups of eight that share a reciprocal seed of 26 bits: 1e-8, not rounding)
    assert np.max(np.abs((top + rest)[live] / f[live] - 1)) < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("n_energies,lo,hi", [(128, 0.1, 30.0), (64, 0.05, 12.0), (256, 0.01, 100.0), (300, 0.5, 20.0), (71, 1.0, 9.0),
                                              (128, 1.0, 3.0e4), (200, 1e-3, 1.0), (63, 0.1, 30.0)],
                         ids=["128", "64-one-half-pass", "256-two-passes", "300", "71-ragged-run", "128-into-the-Wien-tail",
                              "200-Rayleigh-Jeans", "63-general-loop"])
def test_spectrum_on_uniform_energy_grids(capi, n_energies, lo, hi):
    """a UNIFORM grid E_j = E_0 + j dE of >= 64 energies takes the recurrence along the energies
    (k_spectrum.hip planck_runs_uniform: e^-x_(j+1) = e^-x_j e^(-dE s), runs of eight energies per lane).  Against the strict
    kernel's full-precision evaluation (ref python/sim5diskspectrum.py:54-88): 1e-6 of every bin; the same for the grid with
    its last bit(s) disturbed (still uniform to 1e-13: same path) and for a grid that is NOT uniform (one energy moved by 1e-6:
    the general loop), which must agree with the uniform path on the undisturbed bins to 1e-7 (two different loops, one
    26-bit reciprocal seed each); the unpaired instantiation (an asymmetric row range) as well."""
    E = np.linspace(lo, hi, n_energies)
    d_fast = capi.image_desc(160, 96, 0.9, 1.2)
    f = capi.disk_spectrum(d_fast, E)
    s = capi.disk_spectrum(capi.image_desc(160, 96, 0.9, 1.2, strict=True), E)
    assert np.isfinite(f).all() and np.isfinite(s).all() and s.max() > 0
    live = s > 1e-280 * s.max()
    assert not (f[~live] > 1e-270 * s.max()).any()
    err = np.max(np.abs(f[live] / s[live] - 1))
    assert err < 1e-6, (n_energies, lo, hi, err, int(np.argmax(np.abs(f[live] / s[live] - 1))))
    # the general loop on (almost) the same grid: one energy moved, every other bin must agree with the recurrence
    E2 = E.copy(); E2[n_energies // 2] *= 1.0 + 1e-6
    g = capi.disk_spectrum(d_fast, E2)
    keep = live.copy(); keep[n_energies // 2] = False
    assert np.max(np.abs(g[keep] / f[keep] - 1)) < 1e-7
    # a grid made by another expression of the same step (last bits differ): the same path, the same numbers to rounding
    E3 = lo + np.arange(n_energies) * ((hi - lo) / (n_energies - 1))
    h = capi.disk_spectrum(d_fast, E3)
    assert np.max(np.abs(h[live] / f[live] - 1)) < 1e-9
    # the unpaired instantiation and additivity over the rows
    top = capi.disk_spectrum(capi.image_desc(160, 96, 0.9, 1.2, y0=0, y1=37), E)
    rest = capi.disk_spectrum(capi.image_desc(160, 96, 0.9, 1.2, y0=37, y1=96), E)
    top_s = capi.disk_spectrum(capi.image_desc(160, 96, 0.9, 1.2, y0=0, y1=37, strict=True), E)
    lt = top_s > 1e-280 * max(top_s.max(), 1e-300)
    assert np.max(np.abs(top[lt] / top_s[lt] - 1)) < 1e-6
    assert np.max(np.abs((top + rest)[live] / f[live] - 1)) < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fast"])
def test_thick_disk_surface_search(golden, capi, strict):
    """sim5gpu_disk_surface_rays against the reference's Python __find_surface run on the same table."""
    g = golden("py_diskraytrace.npz")
    for ci, (a, inc) in enumerate(g["surf_cases"]):
        s = capi.disk_surface_rays(float(a), math.radians(float(inc)), g["surf_R"], g["surf_H"],
                                   g["surf_alpha"], g["surf_beta"], strict=strict)
        ok = g["surf%d_ok" % ci] == 1
        assert np.array_equal(s["status"] == 1, ok), (ci, int(((s["status"] == 1) != ok).sum()))
        r_ref, m_ref, k_ref = g["surf%d_r" % ci][ok], g["surf%d_m" % ci][ok], g["surf%d_k" % ci][ok]
        # the search stops within `accuracy` = 1e-2 of the surface at a point fixed by its step sequence;
        # identical sequences give identical points
        assert np.max(np.abs(s["r"][ok] / r_ref - 1)) < 1e-6, (ci, np.max(np.abs(s["r"][ok] / r_ref - 1)))
        assert np.max(np.abs(s["m"][ok] - m_ref)) < 1e-6
        assert np.max(np.abs(s["k"][ok] - k_ref) / np.maximum(np.abs(k_ref), 1e-6)) < 1e-5
        # and the points do lie on the tabulated surface to the search accuracy
        R = s["r"][ok] * np.sqrt(1 - s["m"][ok] ** 2); H = s["r"][ok] * s["m"][ok]
        Hd = np.interp(R, g["surf_R"], g["surf_H"])
        assert np.max(np.abs(H - Hd)) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fast"])
def test_thick_disk_surface_search_over_tables_spins_and_fields(golden, capi, strict):
    """sim5gpu_disk_surface_rays against the reference's Python __find_surface over what the six cases above hold fixed
    (oracle/gen_golden_surface.py: 60 jobs, 11 760 rays): tables in equal, logarithmic and growing steps with 2 .. 256 nodes --
    flat, thin, steep, with a bump --, spins 0 .. 0.998, inclinations 8 .. 85 degrees (one job with the observer below the
    disk's surface: no ray arrives), fields of view from 12 to 150 (the wide ones have rays that go through the reference's
    retries).  The same rays succeed, at the same points."""
    g = golden("py_surface_more.npz")
    rays = hits = 0
    worst = {"r": 0.0, "m": 0.0, "k": 0.0}
    for ci, (a, inc, rmax, ti) in enumerate(g["cases"]):
        a = max(float(a), 1e-4)          # the reference's class works with this spin throughout (python/sim5diskraytrace.py:32), as does sim5_amd/diskraytrace.py
        s = capi.disk_surface_rays(float(a), math.radians(float(inc)), g["c%d_R" % ci], g["c%d_H" % ci],
                                   g["c%d_alpha" % ci], g["c%d_beta" % ci], strict=strict)
        ok = g["c%d_ok" % ci] == 1
        assert np.array_equal(s["status"] == 1, ok), (ci, str(g["table_names"][int(ti)]), float(a), float(inc), int(((s["status"] == 1) != ok).sum()))
        rays += ok.size; hits += int(ok.sum())
        if not ok.any():
            continue
        r_ref, m_ref, k_ref = g["c%d_r" % ci][ok], g["c%d_m" % ci][ok], g["c%d_k" % ci][ok]
        worst["r"] = max(worst["r"], float(np.max(np.abs(s["r"][ok] / r_ref - 1))))
        worst["m"] = max(worst["m"], float(np.max(np.abs(s["m"][ok] - m_ref))))
        # (a component of k next to a turning point of the ray is the root of a difference that cancels -- photon_momentum,
        # src/sim5kerr.c:1176-1177 -- and r within 5e-10 moves it by far more than its own size: each component is held to
        # 1e-5 of itself or of 1e-3 of the ray's largest component, whichever is larger)
        kscale = np.maximum(np.abs(k_ref), 1e-3 * np.max(np.abs(k_ref), axis=1, keepdims=True))
        ek = np.abs(s["k"][ok] - k_ref) / kscale
        if ek.max() > worst["k"]:
            j = np.unravel_index(int(np.argmax(ek)), ek.shape)
            worst["k_where"] = "job %d ray %d component %d: k %s reference %s" % (ci, int(np.nonzero(ok)[0][j[0]]), j[1], s["k"][ok][j[0]].tolist(), k_ref[j[0]].tolist())
        worst["k"] = max(worst["k"], float(ek.max()))
    print("surface search, %d jobs, %d rays (%d on the surface), %s: worst r %.1e m %.1e k %.1e" % (
        len(g["cases"]), rays, hits, "strict" if strict else "fast", worst["r"], worst["m"], worst["k"]))
    assert hits > 0.8 * rays
    assert worst["r"] < 1e-6 and worst["m"] < 1e-6 and worst["k"] < 1e-5, worst


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(0.9, 70.0, 20.0), (0.998, 30.0, 12.0), (0.2, 60.0, 40.0)], ids=["a0.9", "a0.998", "a0.2"])
def test_surface_walk_increments_match_full_evaluation(capi, case):
    """The fast variant walks with the addition theorems of sn, cn, dn between full evaluations (s5_geod.hpp, GeodTrack::
    Along); the strict variant evaluates r(P), mu(P) in full, in the reference's operation order, at every sub-step.  The
    stop point is fixed by the sequence of sub-steps, so identical sequences give the same point to rounding: on a
    256 x 256 image (near-critical rays, both ladder depths, rays through the hole included) the two must agree to
    1e-9 except where a height comparison at the accuracy threshold went the other way (a handful of rays)."""
    a, inc, rmax = case
    n = 256
    ax = ((np.arange(n) + .5) / n - .5) * 2 * rmax
    al, be = np.meshgrid(ax, ax)
    tR = np.linspace(2.0, 60.0, 256); tH = 0.25 * (tR - 2.0)
    f = capi.disk_surface_rays(a, math.radians(inc), tR, tH, al.ravel(), be.ravel(), strict=False)
    s = capi.disk_surface_rays(a, math.radians(inc), tR, tH, al.ravel(), be.ravel(), strict=True)
    same = (f["status"] == s["status"])
    assert same.mean() > 0.9995, (1 - same.mean())
    ok = same & (s["status"] == 1)
    assert ok.sum() > 0.8 * n * n
    dr = np.abs(f["r"][ok] / s["r"][ok] - 1); dm = np.abs(f["m"][ok] - s["m"][ok]); dP = np.abs(f["P"][ok] / s["P"][ok] - 1)
    near = (dr < 1e-9) & (dm < 1e-9) & (dP < 1e-9)
    assert near.mean() > 0.999, (1 - near.mean(), np.sort(dr)[-5:])
    # and the others stopped one decision apart, not somewhere else: both lie on the surface to the search accuracy
    R = f["r"][ok] * np.sqrt(1 - f["m"][ok] ** 2); H = f["r"][ok] * f["m"][ok]
    assert np.max(np.abs(H - np.interp(R, tR, tH, right=np.nan))[R < tR[-1]]) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [False, True], ids=["fast", "strict"])
def test_surface_table_of_maximum_size_and_bad_tables(golden, capi, strict):
    """The accepted maximum of 4096 table nodes (2 x 32 KB of LDS next to the ladders: more than the 64 KB a kernel
    gets without asking) in both variants: the same piecewise-linear surface sampled on 4096 nodes that contain the
    256 original ones gives the same intersection points.  Tables whose radii do not ascend are an argument error."""
    g = golden("py_diskraytrace.npz")
    tR, tH = g["surf_R"], g["surf_H"]
    fine = np.unique(np.concatenate([tR, np.exp(np.linspace(np.log(tR[0]), np.log(tR[-1]), 4096 - len(tR) + 64))]))[:4096]
    fine = np.unique(np.concatenate([tR, fine]))
    fine = np.concatenate([tR, np.setdiff1d(fine, tR)[: 4096 - len(tR)]]); fine.sort()
    assert fine.size == 4096 and np.isin(tR, fine).all()
    fH = np.interp(fine, tR, tH)
    a, inc = g["surf_cases"][1]
    base = capi.disk_surface_rays(float(a), math.radians(float(inc)), tR, tH, g["surf_alpha"], g["surf_beta"], strict=strict)
    big = capi.disk_surface_rays(float(a), math.radians(float(inc)), fine, fH, g["surf_alpha"], g["surf_beta"], strict=strict)
    assert np.array_equal(base["status"], big["status"]) and (base["status"] == 1).sum() > 50
    ok = base["status"] == 1
    # same surface to rounding of the interpolation; the walk compares heights with thresholds, so a ray may stop one
    # sub-step apart -- both ends lie within the search accuracy of the surface
    assert np.max(np.abs(big["r"][ok] - base["r"][ok])) < 2e-2 and np.median(np.abs(big["r"][ok] - base["r"][ok])) < 1e-9
    for bad in (np.concatenate([tR[:10], tR[9:]]), tR[::-1].copy()):          # a repeated node; descending radii
        with pytest.raises(capi.Sim5GpuError):
            capi.disk_surface_rays(float(a), math.radians(float(inc)), bad, np.interp(bad, tR, tH), g["surf_alpha"], g["surf_beta"],
                                   strict=strict)
    with pytest.raises(capi.Sim5GpuError):
        capi.disk_surface_rays(float(a), math.radians(float(inc)), np.linspace(1, 50, 4097), np.zeros(4097), g["surf_alpha"],
                               g["surf_beta"], strict=strict)


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [False, True], ids=["fast", "strict"])
def test_surface_height_lookup_on_any_spacing(capi, strict):
    """The walk looks the segment of R up where a table in equal steps has it (k_surface.hip surface_height_guess) and
    falls back to the bisection where that guess and its neighbours do not hold R.  One straight surface H = (R - 2) / 4
    given on 256 equal steps (the guess holds), on 2 and on 3 nodes (too short to guess in), in geometric steps and in
    steps that grow as a power (the neighbours or the bisection decide) is the same surface up to the rounding of the
    interpolation: the same rays hit it, at the same points."""
    a, inc, rmax, n = 0.9, 70.0, 20.0, 128
    ax = ((np.arange(n) + .5) / n - .5) * 2 * rmax
    al, be = np.meshgrid(ax, ax)
    tables = {"equal": np.linspace(2.0, 60.0, 256), "two": np.array([2.0, 60.0]), "three": np.array([2.0, 7.0, 60.0]),
              "geometric": np.geomspace(2.0, 60.0, 200), "power": 2.0 + 58.0 * np.linspace(0, 1, 300) ** 1.7,
              "nearly_equal": np.linspace(2.0, 60.0, 256) + 0.2 * np.sin(np.arange(256.0))}
    tables["nearly_equal"][[0, -1]] = 2.0, 60.0
    res = {k: capi.disk_surface_rays(a, math.radians(inc), t, 0.25 * (t - 2.0), al.ravel(), be.ravel(), strict=strict)
           for k, t in tables.items()}
    base = res["equal"]
    ok = base["status"] == 1
    assert ok.sum() > 0.8 * n * n
    for k, s in res.items():
        assert (s["status"] == base["status"]).mean() > 0.9995, k
        both = ok & (s["status"] == 1)
        d = np.abs(s["r"][both] - base["r"][both])
        assert np.max(d) < 2e-2 and np.median(d) < 1e-9, (k, np.max(d), np.median(d))


@pytest.mark.gpu
def test_thick_disk_image(golden, capi):
    """DiskRaytrace.image() for a disk with a tabulated photosphere (surface search kernel + surface tetrad,
    g-factor and emission angle through the batch calls) against the reference's own Python class run on the
    same model (oracle/gen_golden_py.py: H(R) table, Novikov-Thorne flux and ell, vr = -0.05/sqrt(R))."""
    from sim5_amd.diskraytrace import DiskModel_Surface, DiskRaytrace
    g = golden("py_diskraytrace.npz")
    Ns, rmax = 12, 30.0
    for ci, (a, inc) in enumerate(g["surf_cases"]):
        disk = DiskModel_Surface(10.0, float(a), 0.1, 0.1, g["surf_R"], g["surf_H"], table_vr=g["surf_V"])
        assert disk.fused
        ref = {k: g["thk%d_%s" % (ci, k)] for k in ("flux", "gfactor", "mue", "T", "R", "H", "V")}
        have = np.isfinite(ref["flux"])
        assert have.sum() > 30
        # one kernel (sim5gpu_disk_surface_frame) and the call-by-call path must both reproduce the reference
        for fused in (True, False):
            img = DiskRaytrace(10.0, float(a), 10.0, disk).image(float(inc), rmax, Ns, fused=fused)
            assert np.array_equal(np.isfinite(img["flux"]), have), (ci, fused, int((np.isfinite(img["flux"]) != have).sum()))
            for k, tol in (("R", 1e-6), ("H", 1e-5), ("gfactor", 1e-6), ("T", 1e-6), ("V", 1e-6), ("flux", 1e-5)):
                err = np.max(np.abs(img[k][have] - ref[k][have]) / np.maximum(np.abs(ref[k][have]), 1e-3 * np.abs(ref[k][have]).max()))
                assert err < tol, (ci, fused, k, err)
            assert np.max(np.abs(img["mue"][have] - ref["mue"][have])) < 1e-3          # degrees


@pytest.mark.gpu
def test_thick_disk_image_over_tables_flows_and_disks(golden, capi):
    """DiskRaytrace.image() of a disk with a tabulated photosphere against the reference's own Python class over what the six
    cases above hold fixed (oracle/gen_golden_thick.py, 30 jobs of 12 x 12 pixels): six surface tables (equal, logarithmic and
    growing steps; thin, steep, flaring, with a bump), no / slow / fast radial inflow, spins 0 .. 0.998, inclinations 8 .. 80
    degrees (at 80 most rays end in the disk's flank), masses 10 and 1e6, two accretion rates, fields of view 15 / 30 / 80.  The
    one kernel (sim5gpu_disk_surface_frame) and the call-by-call path: the same pixels lit, every plane within the
    tolerances of test_thick_disk_image."""
    from sim5_amd.diskraytrace import DiskModel_Surface, DiskRaytrace
    g = golden("py_thick_more.npz")
    worst = {}
    for ci, (a, inc, rmax, Ns, mass, mdot, ti, vkind) in enumerate(g["cases"]):
        disk = DiskModel_Surface(float(mass), float(a), float(mdot), 0.1, g["c%d_tR" % ci], g["c%d_tH" % ci], table_vr=g["c%d_tV" % ci])
        ref = {k: g["c%d_%s" % (ci, k)] for k in ("flux", "gfactor", "mue", "T", "R", "H", "V")}
        have = np.isfinite(ref["flux"])
        for fused in (True, False):
            img = DiskRaytrace(float(mass), float(a), 10.0, disk).image(float(inc), float(rmax), int(Ns), fused=fused)
            assert np.array_equal(np.isfinite(img["flux"]), have), (ci, fused, int((np.isfinite(img["flux"]) != have).sum()))
            if not have.any():
                continue
            for k, tol in (("R", 1e-6), ("H", 1e-5), ("gfactor", 1e-6), ("T", 1e-6), ("V", 1e-6), ("flux", 1e-5)):
                scale = np.maximum(np.abs(ref[k][have]), 1e-3 * max(float(np.abs(ref[k][have]).max()), 1e-300))
                err = float(np.max(np.abs(img[k][have] - ref[k][have]) / scale))
                worst[k] = max(worst.get(k, 0.0), err)
                assert err < tol, (ci, fused, k, err, [float(x) for x in g["cases"][ci]])
            emu = float(np.max(np.abs(img["mue"][have] - ref["mue"][have])))
            worst["mue_deg"] = max(worst.get("mue_deg", 0.0), emu)
            assert emu < 1e-3, (ci, fused, emu)          # degrees
    print("thick-disk images against the reference's Python class, %d jobs: worst %s" % (len(g["cases"]), {k: "%.1e" % v for k, v in worst.items()}))

