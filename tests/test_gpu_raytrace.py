"""Step-wise integrator (Verlet) and the polarized / torus whole-job kernels on the GPU."""
import math

import numpy as np
import pytest

import oraclelib as ol
import gen_golden_access as gga
from gpuutil import REL, assert_close, deg2rad

pytestmark = pytest.mark.gpu

# Step counts are integer work decided by float thresholds (ref src/sim5raytrace.c:213,220): every ray's raytrace() call
# count must equal the CPU loop's, in BOTH arithmetic variants.
STEP_MATCH_STRICT = 1.0
STEP_MATCH_FAST = 1.0


def test_prepare_and_single_step(capi, golden):
    g = golden("kat_raytrace_api.npz")
    rtd = capi.raytrace_prepare(g["a"], g["x"], g["k"], g["precision"], g["options"])
    ref0 = np.frombuffer(g["rtd_prepared"].tobytes(), dtype=capi.RAYTRACE_DTYPE)
    assert np.array_equal(rtd["opt_gr"], ref0["opt_gr"])
    for f in ("step_epsilon", "E", "Q", "kt"):
        assert_close(rtd[f], ref0[f], floor=1e-9, what="prepare." + f)
    assert_close(rtd["dk"], ref0["dk"], floor=1e-9, what="prepare.dk")
    # one raytrace() call from the reference's prepared state
    x1, k1, st, rtd1 = capi.raytrace(g["x"], g["k"], g["stepcap"], ref0, nsteps=1)
    ref1 = np.frombuffer(g["rtd_stepped"].tobytes(), dtype=capi.RAYTRACE_DTYPE)
    assert_close(st, g["step"], what="step taken")
    assert_close(x1, g["x1"], floor=1e-6, what="x after one step")
    assert_close(k1, g["k1"], floor=1e-6, what="k after one step")
    assert np.array_equal(rtd1["pass_"], ref1["pass_"])
    assert_close(rtd1["kt"], ref1["kt"], what="kt")
    assert_close(rtd1["dk"], ref1["dk"], floor=1e-6, what="dk after one step")
    err = capi.raytrace_error(g["x1"], g["k1"], ref1)
    assert_close(err, g["carter"], floor=1e-3, what="raytrace_error")


def test_step_sequences_follow_reference(capi, golden):
    """The first 64 raytrace() calls of each recorded ray, state after each block of calls."""
    g = golden("kat_raytrace.npz")
    cases = g["cases"]
    nray = len(cases)
    x = np.stack([g["x0_%d" % i] for i in range(nray)]); k = np.stack([g["k0_%d" % i] for i in range(nray)])
    rtd = capi.raytrace_prepare(cases[:, 0], x, k, cases[:, 5], cases[:, 6].astype(np.int32))
    done = 0
    for upto in (1, 8, 32, 64):
        x, k, st, rtd = capi.raytrace(x, k, 1e9, rtd, nsteps=upto - done)
        done = upto
        for i in range(nray):
            n = int(g["n_%d" % i][0])
            if n < upto:
                continue
            row = g["tr_%d" % i][upto - 1]            # the first 64 steps are stored contiguously
            tol = 1e-9 * upto
            assert np.max(np.abs(x[i] - row[0:4]) / np.maximum(np.abs(row[0:4]), 1.0)) < tol, (i, upto)
            assert np.max(np.abs(k[i] - row[4:8]) / np.maximum(np.abs(row[4:8]), 1e-3)) < 1e3 * tol, (i, upto)
            assert abs(st[i] - row[8]) <= 1e-6 * row[8], (i, upto)


def test_polarized_image(capi, golden):
    g = golden("img_c3_polarized.npz")
    n, a, inc = int(g["n"][0]), float(g["a"][0]), float(g["inc_deg"][0])
    d = capi.image_desc(n, n, a, inc / 180.0 * math.pi, pol_degree=0.1)
    N = n * n
    st = capi.DeviceBuffer(3 * N * 8); chi = capi.DeviceBuffer(N * 8); gg = capi.DeviceBuffer(N * 8); cls = capi.DeviceBuffer(N)
    capi.disk_image_polarized_device(d, st.ptr, chi.ptr, aux={"g": gg.ptr, "cls": cls.ptr})
    capi.synchronize()
    S = st.to_numpy(np.float64, (3, n, n)); CH = chi.to_numpy(np.float64, (n, n)); G = gg.to_numpy(np.float64, (n, n))
    iy, ix = g["iy"], g["ix"]
    ch = CH[iy, ix]
    assert np.array_equal(np.isnan(ch), np.isnan(g["chi"]))
    m = ~np.isnan(ch)
    # chi is an angle in (-pi, pi]: compare on the circle
    dchi = np.angle(np.exp(1j * (ch[m] - g["chi"][m])))
    assert np.max(np.abs(dchi)) < 1e-6
    assert_close(G[iy, ix][m], g["g"][m], what="g")
    I, Q, U = S[0][iy, ix][m], S[1][iy, ix][m], S[2][iy, ix][m]
    assert_close(Q, 0.1 * I * np.cos(2 * g["chi"][m]), floor=1e-6 * I.max(), what="Q")
    assert_close(U, 0.1 * I * np.sin(2 * g["chi"][m]), floor=1e-6 * I.max(), what="U")
    assert np.allclose(np.hypot(S[1], S[2]), 0.1 * S[0], rtol=1e-12, atol=0)       # |P| = delta I everywhere
    # intensity plane equals the unpolarized kernel's F g^4
    o = capi.disk_image(capi.image_desc(n, n, a, inc / 180.0 * math.pi), full=True)
    g2 = o["g"] * o["g"]
    assert_close(S[0], o["flux"] * (g2 * g2), rtol=1e-12, what="I plane vs unpolarized kernel")
    assert np.array_equal(cls.to_numpy(np.uint8, (n, n)), o["cls"])


def test_polarized_image_every_pixel_against_the_live_reference(capi):
    """VERDICT r5 weak 1 (siblings of the flux floor): the C3 image -- 2048^2, a = 0.9, i = 70 deg -- EVERY pixel against the chain
    of reference routines run live on the host (oracle/cpu_driver.c cpu_polarized_rays over oracle/_ref: geodesic_momentum ->
    kerr_metric -> tetrad_azimuthal -> polarization_constant -> polarization_angle_rotation; our restatement where the reference
    library is absent): the same pixels carry an angle; chi within 1e-6 RELATIVE where |chi| > 1e-3 (1e-9 absolute below that),
    compared on the circle; Q and U within 1e-6 of the pixel's OWN polarized intensity delta I (no floor by the image's peak:
    Q = delta I cos 2 chi has zeros, the scale of its error is delta I of that pixel); g within 1e-6."""
    import ctypes as C
    n, a, inc = 2048, 0.9, 70.0
    d = capi.image_desc(n, n, a, deg2rad(inc), pol_degree=0.1)
    N = n * n
    st = capi.DeviceBuffer(3 * N * 8); chi = capi.DeviceBuffer(N * 8); gg = capi.DeviceBuffer(N * 8)
    capi.disk_image_polarized_device(d, st.ptr, chi.ptr, aux={"g": gg.ptr})
    capi.synchronize()
    S = st.to_numpy(np.float64, (3, N)); CH = chi.to_numpy(np.float64, (N,)); G = gg.to_numpy(np.float64, (N,))
    kind = "reference" if ol.have_reference() else "port"
    lib, prefix = (ol.REF_SO, b"") if kind == "reference" else (ol.ORACLE_SO, b"orc_")
    R = ol.Reference() if kind == "reference" else ol.Oracle()
    rmax = R.r_ms(a) + 8.0
    idx = np.arange(n)
    ixg, iyg = np.meshgrid(idx, idx)
    al = np.ascontiguousarray((((ixg + .5) / n - 0.5) * 2.0 * rmax).ravel())
    be = np.ascontiguousarray((((iyg + .5) / n - 0.5) * 2.0 * rmax * (float(n) / float(n))).ravel())
    ref_chi = np.zeros(N); ref_r = np.zeros(N); ref_g = np.zeros(N); wp = np.zeros((N, 2))
    drv = C.CDLL(ol.DRIVER_SO)
    rc = drv.cpu_polarized_rays(lib.encode(), prefix, C.c_double(a), C.c_double(deg2rad(inc)), C.c_double(-1.0), C.c_int(N),
                                C.c_void_p(al.ctypes.data), C.c_void_p(be.ctypes.data), C.c_void_p(ref_chi.ctypes.data),
                                C.c_void_p(ref_r.ctypes.data), C.c_void_p(ref_g.ctypes.data), C.c_void_p(wp.ctypes.data))
    assert rc == 0
    assert np.array_equal(np.isnan(CH), np.isnan(ref_chi)), "pixels with a polarization angle differ from the %s's: %d" % (kind, int((np.isnan(CH) != np.isnan(ref_chi)).sum()))
    m = ~np.isnan(ref_chi)
    assert m.sum() > 1.5e6
    dchi = np.abs(np.angle(np.exp(1j * (CH[m] - ref_chi[m]))))
    big = np.abs(ref_chi[m]) > 1e-3
    worst_rel = float(np.max(dchi[big] / np.abs(ref_chi[m][big])))
    worst_abs_small = float(dchi[~big].max()) if (~big).any() else 0.0
    assert worst_rel < 1e-6 and worst_abs_small < 1e-9, (worst_rel, worst_abs_small)
    assert_close(G[m], ref_g[m], what="g, every lit pixel")
    I = S[0][m]
    lit = I > 0                                                   # (hits inside the zero-flux band carry an angle and no intensity)
    eq = np.max(np.abs(S[1][m][lit] - 0.1 * I[lit] * np.cos(2 * ref_chi[m][lit])) / (0.1 * I[lit]))
    eu = np.max(np.abs(S[2][m][lit] - 0.1 * I[lit] * np.sin(2 * ref_chi[m][lit])) / (0.1 * I[lit]))
    assert eq < 1e-6 and eu < 1e-6, (eq, eu)
    print("C3 polarized, all %d angle-carrying pixels vs the live %s: chi rel %.2e (|chi| > 1e-3), abs %.2e below; Q %.2e, U %.2e of the pixel's delta I"
          % (int(m.sum()), kind, worst_rel, worst_abs_small, eq, eu))


def test_polarized_mirrored_pairs_give_the_plain_image(capi):
    """A symmetric row range is traced by the pairing kernel (pixel + its mirror image in beta per lane), any other by the
    plain one: Stokes I, Q, U, the angle and the aux planes must agree bit for bit; odd height included."""
    def pol(nx, ny, y0, y1, a, inc):
        d = capi.image_desc(nx, ny, a, deg2rad(inc), y0=y0, y1=y1, pol_degree=0.1)
        rows = y1 - y0
        N = rows * nx
        st = capi.DeviceBuffer(3 * N * 8); chi = capi.DeviceBuffer(N * 8); gg = capi.DeviceBuffer(N * 8); cls = capi.DeviceBuffer(N)
        capi.disk_image_polarized_device(d, st.ptr, chi.ptr, aux={"g": gg.ptr, "cls": cls.ptr})
        capi.synchronize()
        return (st.to_numpy(np.float64, (3, rows, nx)), chi.to_numpy(np.float64, (rows, nx)),
                gg.to_numpy(np.float64, (rows, nx)), cls.to_numpy(np.uint8, (rows, nx)))
    for (nx, ny, a, inc) in [(200, 128, 0.9, 60.0), (131, 77, 0.998, 75.0)]:
        S, CH, G, CL = pol(nx, ny, 0, ny, a, inc)
        cut = ny // 2 + 3
        S1, CH1, G1, CL1 = pol(nx, ny, 0, cut, a, inc)
        S2, CH2, G2, CL2 = pol(nx, ny, cut, ny, a, inc)
        assert np.array_equal(S, np.concatenate([S1, S2], axis=1), equal_nan=True)
        assert np.array_equal(CH, np.concatenate([CH1, CH2]), equal_nan=True)
        assert np.array_equal(G, np.concatenate([G1, G2])) and np.array_equal(CL, np.concatenate([CL1, CL2]))
        assert (S[0] > 0).sum() > 0.3 * nx * ny


def torus_desc(capi, n, a, inc_deg, **kw):
    img = capi.image_desc(n, n, a, inc_deg / 180.0 * math.pi)
    d = capi.TorusDesc(img=img, r0=kw.get("r0", 100.0), dl_max=kw.get("dl_max", 1e9),
                       precision=kw.get("precision", 1.0), options=kw.get("options", 0),
                       max_steps=kw.get("max_steps", 20000), max_error=kw.get("max_error", 1e-2), r_stop_in=kw.get("r_stop_in", 1.05), r_stop_out=1.01,
                       shape=kw.get("shape", 0), torus_r=kw.get("torus_r", 8.0), torus_w=kw.get("torus_w", 2.0),
                       torus_l=kw.get("torus_l", 3.5), emis0=kw.get("emis0", 1.0), absorb0=kw.get("absorb0", 0.0))
    return d


def run_torus(capi, d, full=False):
    n = d.img.nx * (d.img.y1 - d.img.y0)
    st = capi.DeviceBuffer(n * 40); steps = capi.DeviceBuffer(n * 4); xe = capi.DeviceBuffer(n * 32)
    ce = capi.DeviceBuffer(n * 8); me = capi.DeviceBuffer(n * 4); ke = capi.DeviceBuffer(n * 32)
    # sentinels: a ray the job does not write keeps them
    st.fill(0xff); steps.fill(0xff)
    capi.torus_image_device(d, st.ptr, aux={"steps": steps.ptr, "x_end": xe.ptr, "carter_error": ce.ptr,
                                            "max_step_error": me.ptr, "k_end": ke.ptr})
    capi.synchronize()
    out = (st.to_numpy(np.float64, (n, 5)), steps.to_numpy(np.int32, (n,)), xe.to_numpy(np.float64, (n, 4)),
           ce.to_numpy(np.float64, (n,)), me.to_numpy(np.float32, (n,)))
    return out + (ke.to_numpy(np.float64, (n, 4)),) if full else out


FIELDS = (("t_end", "x_end", 0, 1.0), ("r_end", "x_end", 1, 0.0), ("cos(theta)_end", "x_end", 2, 1e-2), ("phi_end", "x_end", 3, 1.0),
          ("k_end[0]", "k_end", 0, 0.0), ("k_end[1]", "k_end", 1, 1e-2), ("k_end[2]", "k_end", 2, 1e-4), ("k_end[3]", "k_end", 3, 1e-4))


def ray_errors(got, ref):
    """Per-ray relative differences of two results of the same rays (dicts with x_end, k_end, I, tau): t and phi are sums
    over the whole ray (|t| ~ 200, |phi| up to a few turns: floor 1), r and k^t are O(1..100) (no floor), cos(theta), k^r,
    k^theta, k^phi pass through zero (floors 1e-2, 1e-2, 1e-4, 1e-4: k^theta, k^phi ~ 1e-2 at r ~ 100); I and tau against
    1e-6 of the set's peak.  Returns {field: array}."""
    out = {}
    for name, key, c, floor in FIELDS:
        out[name] = np.abs(got[key][:, c] - ref[key][:, c]) / np.maximum(np.abs(ref[key][:, c]), floor)
    out["Stokes I"] = np.abs(got["I"] - ref["I"]) / np.maximum(np.abs(ref["I"]), 1e-6 * float(np.abs(ref["I"]).max()))
    out["tau"] = np.abs(got["tau"] - ref["tau"]) / np.maximum(np.abs(ref["tau"]), 1e-6 * max(float(np.abs(ref["tau"]).max()), 1e-300))
    return out


def oracle_sensitivity(lib, prefix, a, inc_rad, alpha, beta, base, **job):
    """Conditioning probe: how far does the CHECKER's own result of these rays move when its start state changes in the
    last bit?  Each of cos(theta), k^r, k^theta, k^phi at the start by +-1 ulp before raytrace_prepare()
    (oracle/cpu_driver.c:cpu_torus_rays_perturbed), the largest move per ray in the metric of ray_errors(); a ray whose
    step count changes under such a perturbation sits on a threshold of the step control (infinite sensitivity)."""
    kap = np.zeros(len(alpha))
    for comp in (2, 5, 6, 7):
        for sgn in (1, -1):
            u = [0] * 8
            u[comp] = sgn
            o = gga.torus_rays(lib, prefix, a, inc_rad, alpha, beta, ulps=u, **job)
            e = np.max(np.stack(list(ray_errors(o, base).values())), axis=0)
            e[o["steps"] != base["steps"]] = np.inf
            kap = np.maximum(kap, e)
    return kap


# every comparison of this module, by tag: how many rays it held to 1e-6 and how many of them passed through the one
# exemption (a ray above 1e-6 whose difference the reference's own +-1 ulp sensitivity covers): written to
# gpurun_out/torus_exemptions.json by the last test of the module, so that the count survives `pytest -q`
EXEMPTIONS = {}


def compare_rays(tag, S, steps, xe, ke, ref, need_same, probe=None):
    """GPU rays against the CPU integration of the same rays (dict of oracle/cpu_driver.c:cpu_torus_rays arrays):
    step counts, and for rays with identical counts the end point (all four coordinates), the end momentum and
    the transfer integrals I and tau within 1e-6 -- in BOTH arithmetic variants, every ray.

    The one admissible exception is a ray that is ill-conditioned in the REFERENCE itself: a photon that winds around
    the photon orbit amplifies a last-bit difference exponentially, for any two implementations.  It is not excused by
    a looser bar but by a measurement: `probe(indices)` reruns the CHECKER on those rays with its start state moved by
    +-1 ulp (oracle_sensitivity) and the ray passes only if the checker's own result moves by at least as much as the
    GPU's differs.  Such rays are counted and printed; more than 0.1 % of a set fails."""
    EXEMPTIONS.setdefault(tag, {"rays": int(steps.size), "above_1e-6": 0, "excused_by_the_reference_own_sensitivity": 0})
    same = steps == ref["steps"]
    print("%s: %d of %d rays with identical step counts (the others differ by %s steps)" % (
        tag, int(same.sum()), same.size, sorted(set((steps - ref["steps"])[~same].tolist()))[:8]))
    assert same.sum() >= need_same * same.size, (tag, int(same.sum()), same.size)
    assert np.array_equal(steps == 0, ref["steps"] == 0)
    m = same & (steps > 0)
    errs = ray_errors({"x_end": xe, "k_end": ke, "I": S[:, 0], "tau": S[:, 4]}, ref)
    print("   worst relative differences:", {k: "%.1e" % (float(v[m].max()) if m.any() else 0.0) for k, v in errs.items()})
    tot = np.max(np.stack(list(errs.values())), axis=0)
    over = np.nonzero(m & (tot > REL))[0]
    if over.size:
        assert probe is not None, "%s: %d rays above 1e-6 (worst %.3e) and no conditioning probe" % (tag, over.size, tot[over].max())
        assert over.size <= 1e-3 * max(int(m.sum()), 1000), "%s: %d of %d rays above 1e-6" % (tag, over.size, int(m.sum()))
        kap = probe(over)
        print("   %d ray(s) above 1e-6: difference %s, the checker's own +-1 ulp sensitivity %s" % (
            over.size, ["%.1e" % v for v in tot[over][:8]], ["%.1e" % v for v in kap[:8]]))
        EXEMPTIONS[tag].update({"above_1e-6": int(over.size), "excused_by_the_reference_own_sensitivity": int((tot[over] <= kap).sum()),
                                "worst_difference": float(tot[over].max()), "smallest_sensitivity_of_an_excused_ray": float(kap.min())})
        assert (tot[over] <= kap).all(), "%s: rays above 1e-6 that are well-conditioned in the checker: %s" % (
            tag, [(int(i), float(t), float(k)) for i, t, k in zip(over, tot[over], kap) if t > k][:8])
    assert (S[:, 1:4] == 0).all()
    return same


def test_torus_flat_space_uniform_sphere(capi):
    """RTOPT_FLAT, static uniform sphere of radius R, no absorption: I = emis0 * chord length."""
    n, R = 48, 6.0
    d = torus_desc(capi, n, 0.5, 60.0, options=1, shape=1, torus_w=R, torus_l=0.0, r0=40.0, dl_max=0.02,
                   max_steps=200000, max_error=1e30, r_stop_in=1e-3)
    d.img.rmax = 8.0
    S, steps, xe, ce, me = run_torus(capi, d)
    c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * 8.0
    b = np.hypot(c[None, :], c[:, None]).ravel()               # impact parameter = distance from the axis of sight
    chord = np.where(b < R, 2.0 * np.sqrt(np.maximum(R * R - b * b, 0.0)), 0.0)
    err = np.abs(S[:, 0] - chord)
    assert (S[:, 0][b > R + 0.05] == 0).all()
    # rays that pass within ~2 of the origin hit the coordinate singularity of spherical coordinates
    # (1/r and cot(theta) terms of the flat connection): the reference's integrator is not meant for
    # that region (in Kerr no ray gets below the horizon), so the analytic check is made outside it
    sel = (b > 2.1) & (np.abs(b - R) > 0.3)
    assert sel.sum() > 500
    assert np.max(err[sel]) < 0.03, np.max(err[sel])
    assert (steps[sel] > 3000).all() and (xe[sel, 1] > 40.0).all()


@pytest.mark.parametrize("strict", [False, True], ids=["fast", "strict"])
@pytest.mark.parametrize("absorb0", [0.0, 0.3], ids=["thin", "absorbing"])
def test_torus_kernel_matches_cpu_integration(capi, strict, absorb0):
    """Kerr, a = 0.9, a 24 x 24 image: per ray the step count, the end point (t, r, cos theta, phi), the end momentum
    and the transfer integrals I and tau against the oracle's raytrace() loop with the same per-step accumulation done
    on the CPU (oracle/cpu_driver.c:cpu_torus_rays); with and without absorption (the exp(-tau) branch)."""
    n, a, inc, r0 = 24, 0.9, 70.0, 100.0
    d = torus_desc(capi, n, a, inc, r0=r0, absorb0=absorb0)
    if strict:
        d.img.flags = 1
    S, steps, xe, ce, me, ke = run_torus(capi, d, full=True)
    rmax = ol.Oracle().r_ms(a) + 8.0
    c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * rmax
    al, be = np.tile(c, n), np.repeat(c, n)
    ref = gga.torus_rays(ol.ORACLE_SO, "orc_", a, deg2rad(inc), al, be, r0=r0, absorb0=absorb0)

    def probe(idx):
        sub = {k: v[idx] for k, v in ref.items()}
        return oracle_sensitivity(ol.ORACLE_SO, "orc_", a, deg2rad(inc), al[idx], be[idx], sub, r0=r0, absorb0=absorb0)
    compare_rays("24x24 %s absorb0=%g" % ("strict" if strict else "fast", absorb0), S, steps, xe, ke, ref,
                 STEP_MATCH_STRICT if strict else STEP_MATCH_FAST, probe)
    assert (S[:, 0] >= 0).all() and S[:, 0].max() > 1.0
    if absorb0 > 0:
        assert S[:, 4].max() > 1.0            # optically thick lines of sight are in the sample


def test_random_torus_jobs(capi):
    """25 random step-wise jobs (spin 0.01 ... 0.998, inclination 10 ... 85 deg, r0 40 ... 200, precision 1 ... 0.03, images of 6^2
    ... 27^2 rays, torus size, with and without absorption; tests/tools/fuzz_torus.py runs the open-ended version): the step
    count of EVERY ray equals the CPU checker's raytrace() loop in both variants; end point, end momentum, Stokes I and tau
    within 1e-6 in both variants (compare_rays: a ray may exceed it only if the checker's own result moves as much under a
    +-1 ulp change of its start state).
    Rays with alpha = 0 exactly (central column of an odd-sized image; left out of everything until round 5): l = 0, and whether
    the reference starts such a ray at all is decided by the roundings of its x87 polar roots, which the device now reproduces
    (s5_x87.hpp) -- so they are IN the step-count comparison of both variants.  Their VALUES are not asserted, only counted and
    printed (how many above 1e-6, the worst): an l = 0 ray runs INTO the polar axis, the step control hits its floor dl = 1e-3 (ref
    src/sim5raytrace.c:166) and creeps up to sin(theta) ~ 1e-5, where one ulp of m = cos(theta) is 1e-6 of sin^2(theta):
    last-bit differences of the state grow by 1e2 - 1e4 per step (tests/tools/torus_diverge.py: k^phi 6e-12 -> 2e-5 over the
    last ten steps before the axis).  The strict variant stays within 2e-8 on most of them (its operations are the
    reference's; acos / cos of the device's libm differ from glibc's in the last bit now and then -- one ray of the 25 jobs
    then ends elsewhere altogether), the fast one within 2e-4."""
    rng = np.random.default_rng(2027)
    axis_rays, axis_over, axis_worst = {True: 0, False: 0}, {True: 0, False: 0}, {True: 0.0, False: 0.0}
    for case in range(25):
        a = float(rng.choice([0.1, 0.3, 0.9, 0.998, rng.uniform(0.01, 0.99)]))
        inc = float(rng.uniform(10.0, 85.0))
        n = int(rng.integers(6, 28))
        r0 = float(rng.uniform(40.0, 200.0))
        prec = float(rng.choice([1.0, 1.0, 0.3, 0.1, 0.03]))
        absorb0 = float(rng.choice([0.0, 0.3]))
        tr, tw = float(rng.uniform(5.0, 12.0)), float(rng.uniform(1.0, 3.0))
        what = (case, a, inc, n, r0, prec, absorb0, tr, tw)
        rmax = ol.Oracle().r_ms(a) + 8.0
        c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * rmax
        al, be = np.tile(c, n), np.repeat(c, n)
        job = dict(r0=r0, precision=prec, absorb0=absorb0, torus_r=tr, torus_w=tw)
        ref = gga.torus_rays(ol.ORACLE_SO, "orc_", a, deg2rad(inc), al, be, **job)
        off_axis = np.nonzero(al != 0.0)[0]
        for strict in (True, False):
            regular = off_axis
            sub = {k: v[regular] for k, v in ref.items()}

            def probe(idx):
                return oracle_sensitivity(ol.ORACLE_SO, "orc_", a, deg2rad(inc), al[regular][idx], be[regular][idx],
                                          {k: v[idx] for k, v in sub.items()}, **job)
            d = torus_desc(capi, n, a, inc, **job)
            if strict:
                d.img.flags = 1
            S, steps, xe, ce, me, ke = run_torus(capi, d, full=True)
            assert np.array_equal(steps, ref["steps"]), what + (strict,)             # every ray, alpha = 0 included
            compare_rays("random job %d %s" % (case, "strict" if strict else "fast"), S[regular], steps[regular], xe[regular],
                         ke[regular], sub, 1.0, probe)
            on_axis = np.nonzero((al == 0.0) & (steps > 0))[0]
            if on_axis.size:
                e = ray_errors({"x_end": xe[on_axis], "k_end": ke[on_axis], "I": S[on_axis, 0], "tau": S[on_axis, 4]},
                               {k: v[on_axis] for k, v in ref.items()})
                w = np.max(np.stack(list(e.values())), axis=0)
                axis_rays[strict] += on_axis.size; axis_over[strict] += int((w > REL).sum()); axis_worst[strict] = max(axis_worst[strict], float(w.max()))
    for strict in (True, False):
        print("%s variant, rays into the polar axis (alpha = 0): %d, of them above 1e-6: %d, worst difference %.1e (not asserted)" % (
            "strict" if strict else "fast", axis_rays[strict], axis_over[strict], axis_worst[strict]))


@pytest.mark.parametrize("strict", [False, True], ids=["fast", "strict"])
def test_c4_full_size(capi, golden, strict):
    """BASELINE.json configs[3] at its full size: 1024 x 1024 rays through the torus (the real job: 2 048 persistent
    waves, cursor refill, ~5.4e8 raytrace() calls).  Every ray: written exactly once (sentinels gone, the counters
    of the job add up), started (r0 = 100 is outside every pericentre of this image), ended for one of the three
    reasons of the stop rule, Carter constant conserved to the level of the reference on the same rays.  Every 16th
    pixel of the SAME grid (4 096 rays): step counts, end point, end momentum, I and tau against the unmodified
    reference's raytrace() loop (golden torus_c4.npz), without and with absorption."""
    g = golden("torus_c4.npz")
    n, a, inc, r0 = int(g["n"][0]), float(g["a"][0]), float(g["inc_deg"][0]), 100.0
    rbh = 1.0 + math.sqrt(1.0 - a * a)
    for tag, absorb0 in (("thin", 0.0), ("absorb", 0.3)):
        d = torus_desc(capi, n, a, inc, r0=r0, absorb0=absorb0)
        if strict:
            d.img.flags = 1
        S, steps, xe, ce, me, ke = run_torus(capi, d, full=True)
        N = n * n
        assert not np.isnan(S).any(), "%d rays were never written" % int(np.isnan(S[:, 0]).sum())
        assert (steps >= 0).all() and (steps < 20000).all()
        # rays the start-up rejects (here 134 rays near beta = 0 whose photon_momentum() at r0 is NaN in the reference,
        # ref src/sim5kerr.c:1183-1188): the CPU loop rejects exactly the same ones
        unstarted = np.nonzero(steps == 0)[0]
        print("%d of %d rays not started" % (unstarted.size, N))
        assert unstarted.size < 256
        if unstarted.size and tag == "thin":
            rmax = ol.Oracle().r_ms(a) + 8.0
            c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * rmax
            u = unstarted[:256]
            o = gga.torus_rays(ol.ORACLE_SO, "orc_", a, deg2rad(inc), c[u % n], c[u // n], r0=r0)
            assert (o["steps"] == 0).all(), (u[o["steps"] != 0][:8], o["steps"][o["steps"] != 0][:8])
            assert (S[unstarted] == 0).all()
        started = steps > 0
        inside = started & (xe[:, 1] > 1.05 * rbh) & (xe[:, 1] < 1.01 * r0)
        assert (me[inside] > 1e-2).all()                 # a ray that stopped inside the domain stopped on its error
        assert np.isfinite(ce[started]).all() and np.isfinite(xe[started]).all() and np.isfinite(ke[started]).all()
        assert (S[:, 0] >= 0).all() and (S[:, 4] >= 0).all()
        sel = g["thin_iy"].astype(np.int64) * n + g["thin_ix"]
        ref = {k: g["thin_" + k] for k in ("steps", "x_end", "k_end", "carter", "max_step_error")}
        ref["I"] = g[tag + "_I"]; ref["tau"] = g[tag + "_tau"]
        name = "C4 1024^2 %s %s" % ("strict" if strict else "fast", tag)
        rmax = ol.Oracle().r_ms(a) + 8.0
        cc = ((np.arange(n) + .5) / n - 0.5) * 2.0 * rmax

        def probe(idx, absorb0=absorb0, ref=ref):
            # the restatement is bit-identical to the reference on these rays (tests/test_oracle_golden.py), so its
            # sensitivity is the reference's
            al, be = cc[g["thin_ix"][idx]], cc[g["thin_iy"][idx]]
            return oracle_sensitivity(ol.ORACLE_SO, "orc_", a, deg2rad(inc), al, be, {k: v[idx] for k, v in ref.items()},
                                      r0=r0, absorb0=absorb0)
        same = compare_rays(name, S[sel], steps[sel], xe[sel], ke[sel], ref, STEP_MATCH_STRICT if strict else STEP_MATCH_FAST, probe)
        # conservation: the Carter-constant error of the GPU rays is the reference's on the same rays
        assert_close(ce[sel][same], ref["carter"][same], floor=1e-3, rtol=1e-3, what=name + " raytrace_error")
        print("%s: %.1f steps/ray, %.3g raytrace() calls, I max %.4g, |dQ/Q| median %.2e, 99.9%% %.2e, max %.2e; rays stopped by "
              "their error inside the domain: %d" % (name, steps.mean(), float(steps.sum()), S[:, 0].max(),
                                                     np.median(ce[started]), np.quantile(ce[started], 0.999), ce[started].max(), int(inside.sum())))
        assert np.median(ce[started]) < 2.0 * np.median(ref["carter"])


@pytest.mark.parametrize("strict", [False, True], ids=["fast", "strict"])
def test_c4_grid_fine_precision(capi, golden, strict):
    """The C4 view as a 16 x 16 image at precision 0.01 (~4 100 steps per ray) against the reference's loop."""
    g = golden("torus_c4.npz")
    n, a, inc = int(g["n"][0]), float(g["a"][0]), float(g["inc_deg"][0])
    d = torus_desc(capi, 16, a, inc, precision=0.01, max_steps=50000)
    if strict:
        d.img.flags = 1
    S, steps, xe, ce, me, ke = run_torus(capi, d, full=True)
    ref = {k: g["fine_" + k] for k in ("steps", "x_end", "k_end", "I", "tau")}
    compare_rays("C4 grid, precision 0.01, %s" % ("strict" if strict else "fast"), S, steps, xe, ke, ref,
                 STEP_MATCH_STRICT if strict else STEP_MATCH_FAST)


@pytest.mark.parametrize("strict", [False, True], ids=["fast", "strict"])
def test_polarized_parameter_sweep(capi, strict):
    """Polarization angle and g over a seeded sweep of spins / inclinations, GPU kernel against the CPU
    oracle's recipe (oracle/cpu_driver.c:cpu_polarized_rays run on our restatement) on the same rays."""
    import ctypes as C
    drv = C.CDLL(ol.DRIVER_SO)
    D, I, VP = C.c_double, C.c_int, C.c_void_p
    drv.cpu_polarized_rays.argtypes = [C.c_char_p, C.c_char_p, D, D, D, I, VP, VP, VP, VP, VP, VP]
    drv.cpu_polarized_rays.restype = I
    rng = np.random.default_rng(5)
    n = 96
    for a, inc in [(0.0, 20.0), (0.5, 45.0), (0.998, 80.0)] + [(float(rng.uniform(0, 0.999)), float(rng.uniform(5, 85))) for _ in range(5)]:
        d = capi.image_desc(n, n, a, deg2rad(inc), pol_degree=0.1, strict=strict)
        N = n * n
        st = capi.DeviceBuffer(3 * N * 8); chi = capi.DeviceBuffer(N * 8); gg = capi.DeviceBuffer(N * 8)
        capi.disk_image_polarized_device(d, st.ptr, chi.ptr, aux={"g": gg.ptr})
        capi.synchronize()
        CH = chi.to_numpy(np.float64, (N,)); G = gg.to_numpy(np.float64, (N,))
        rmax = ol.Oracle().r_ms(a) + 8.0
        c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * rmax
        al = np.ascontiguousarray(np.tile(c, n)); be = np.ascontiguousarray(np.repeat(c, n))
        rchi = np.zeros(N); rr = np.zeros(N); rg = np.zeros(N); rwp = np.zeros((N, 2))
        rc = drv.cpu_polarized_rays(ol.ORACLE_SO.encode(), b"orc_", a, deg2rad(inc), -1.0, N, al.ctypes.data, be.ctypes.data,
                                    rchi.ctypes.data, rr.ctypes.data, rg.ctypes.data, rwp.ctypes.data)
        assert rc == 0
        assert np.array_equal(np.isnan(CH), np.isnan(rchi)), (a, inc, int((np.isnan(CH) != np.isnan(rchi)).sum()))
        m = ~np.isnan(rchi)
        assert m.sum() > 1000
        dchi = np.angle(np.exp(1j * (CH[m] - rchi[m])))
        assert np.max(np.abs(dchi)) < 1e-6, (a, inc, float(np.max(np.abs(dchi))))
        assert_close(G[m], rg[m], what="g a=%g i=%g" % (a, inc))


@pytest.mark.parametrize("strict", [False, True], ids=["fast", "strict"])
def test_torus_kernel_parameter_sweep(capi, strict):
    """Step counts and end points of the march kernel against the oracle's raytrace() loop for other spins,
    inclinations and precisions than the C4 configuration, both arithmetic variants."""
    n, r0 = 16, 100.0
    orc = ol.Oracle()
    for a, inc, prec in [(0.2, 30.0, 1.0), (0.998, 80.0, 1.0), (0.6, 55.0, 0.1), (0.9, 70.0, 0.01)]:
        d = torus_desc(capi, n, a, inc, r0=r0, precision=prec, max_steps=50000)
        if strict:
            d.img.flags = 1
        S, steps, xe, ce, me = run_torus(capi, d)
        rmax = orc.r_ms(a) + 8.0
        c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * rmax
        cases = [(a, inc / 180.0 * math.pi, c[ix], c[iy], r0, prec, 0) for iy in range(n) for ix in range(n)]
        res = gga.verlet_traces(ol.ORACLE_SO, "orc_", cases, 50000)
        same = 0
        for i, (m, tr, xs, ks, car) in enumerate(res):
            if m <= 0:
                assert steps[i] == 0
                continue
            same += int(m == steps[i])
            if m == steps[i]:
                assert abs(xe[i, 1] - tr[m - 1, 1]) <= 1e-6 * max(1.0, abs(tr[m - 1, 1])), (a, inc, prec, i)
        print("torus sweep a=%g i=%g precision=%g %s: %d of %d rays with identical step counts" % (a, inc, prec, "strict" if strict else "fast", same, len(res)))
        assert same >= (STEP_MATCH_STRICT if strict else STEP_MATCH_FAST) * len(res), (a, inc, prec, strict, same, len(res))
        assert np.isfinite(S[:, 0]).all()


def test_workspaces_can_be_released(capi):
    """sim5gpu_release_workspaces gives the grow-only workspaces of the torus and surface jobs back; the next job allocates
    again and produces the same result."""
    d = torus_desc(capi, 32, 0.9, 70.0)
    a = run_torus(capi, d)
    tR = np.linspace(2.0, 40.0, 64); tH = 0.2 * (tR - 2.0)
    c = ((np.arange(24) + .5) / 24 - 0.5) * 40.0
    s1 = capi.disk_surface_rays(0.9, deg2rad(70.0), tR, tH, np.tile(c, 24), np.repeat(c, 24))
    freed = capi.release_workspaces()
    assert freed > 32 * 32 * 100
    assert capi.release_workspaces() == 0
    b = run_torus(capi, d)
    s2 = capi.disk_surface_rays(0.9, deg2rad(70.0), tR, tH, np.tile(c, 24), np.repeat(c, 24))
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)
    for k in s1:
        assert np.array_equal(s1[k], s2[k], equal_nan=True)


def _c4_rows_on_the_cpu(args):
    n, rows, absorb0 = args
    c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * (ol.Oracle().r_ms(0.998) + 8.0)
    o = gga.torus_rays(ol.ORACLE_SO, "orc_", 0.998, deg2rad(70.0), np.tile(c, len(rows)), np.repeat(c[rows], n), r0=100.0, absorb0=absorb0)
    return rows[0], {k: o[k] for k in ("steps", "x_end", "k_end", "I", "tau")}


def test_c4_every_ray_of_a_256_image(capi):
    """The C4 job at 256 x 256 with EVERY ray held to the CPU checker's raytrace() loop (the 1024^2 job the same way:
    tests/tools/c4_all_rays.py, profiles/r06_c4_all_rays.json -- the golden sample of test_c4_full_size is every 16th pixel and
    never sees the two columns next to the image's vertical axis).  What holds: call counts identical on every ray in both
    variants; the strict variant within 1e-6 on every ray; the fast variant within 1e-6 on every ray EXCEPT a handful in the two
    columns next to alpha = 0 (|alpha| = half a pixel: the ray passes the polar axis at ~1e-3 rad, where the reference's own
    integration amplifies a last-bit difference a few hundred times and may then take the other side of one of its accept / fall-back
    decisions); Stokes I within 1e-6 on every ray in both.  The smaller the pixel the closer such rays pass: at 1024^2 they are
    ~300 of 1 048 576 in the fast variant (a few ending AT the pole, where phi and k^phi are singular: 0.14) and 12 in the strict one."""
    from multiprocessing import Pool
    import os
    n = 256
    blocks = [np.arange(r, min(r + 16, n)) for r in range(0, n, 16)]
    with Pool(min(16, os.cpu_count() or 1)) as pool:
        parts = dict(pool.map(_c4_rows_on_the_cpu, [(n, b, 0.0) for b in blocks]))
    ref = {k: np.concatenate([parts[b[0]][k] for b in blocks]) for k in ("steps", "x_end", "k_end", "I", "tau")}
    for strict in (True, False):
        d = torus_desc(capi, n, 0.998, 70.0, r0=100.0)
        if strict:
            d.img.flags = 1
        S, steps, xe, ce, me, ke = run_torus(capi, d, full=True)
        assert np.array_equal(steps, ref["steps"]), int((steps != ref["steps"]).sum())
        m = steps > 0
        errs = ray_errors({"x_end": xe, "k_end": ke, "I": S[:, 0], "tau": S[:, 4]}, ref)
        tot = np.max(np.stack([np.nan_to_num(v, nan=0.0) for v in errs.values()]), axis=0)
        over = np.nonzero(m & (tot > REL))[0]
        print("C4 256^2, every ray, %s: %d rays above 1e-6 (worst %.1e), columns %s; worst Stokes I %.1e" % (
            "strict" if strict else "fast", over.size, float(tot[m].max()), sorted(set((over % n).tolist())), float(np.nan_to_num(errs["Stokes I"][m]).max())))
        assert np.nan_to_num(errs["Stokes I"][m]).max() < REL
        if strict:
            assert over.size == 0, (over[:8], tot[over][:8])
        else:
            assert over.size <= 12 and all(abs((i % n) - (n - 1) / 2.0) <= 2.0 for i in over), [(int(i % n), int(i // n), float(tot[i])) for i in over[:12]]
            assert tot[over].max() < 1e-4 if over.size else True


def test_zz_exemption_record(capi):
    """Not a comparison: writes what the comparisons above recorded -- per set the rays held to 1e-6 and how many of them used
    the ill-conditioned-ray exemption of compare_rays -- to gpurun_out/torus_exemptions.json (committed under profiles/), and
    holds the total of excused rays under 0.1 %."""
    import json, os
    total = sum(v["rays"] for v in EXEMPTIONS.values())
    excused = sum(v["excused_by_the_reference_own_sensitivity"] for v in EXEMPTIONS.values())
    rec = {"sets": len(EXEMPTIONS), "rays_compared": total, "rays_excused": excused,
           "sets_with_excused_rays": {k: v for k, v in EXEMPTIONS.items() if v["above_1e-6"]}}
    print("step-wise comparisons: %d sets, %d rays, %d excused by the reference's own sensitivity" % (len(EXEMPTIONS), total, excused))
    outdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(outdir):
        with open(os.path.join(outdir, "torus_exemptions.json"), "w") as fh:
            json.dump(rec, fh, indent=1)
    assert total == 0 or excused <= 1e-3 * total, rec
