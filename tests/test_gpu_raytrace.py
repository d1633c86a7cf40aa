"""Step-wise integrator (Verlet) and the polarized / torus whole-job kernels on the GPU."""
import math

import numpy as np
import pytest

import oraclelib as ol
import gen_golden_access as gga
from gpuutil import REL, assert_close

pytestmark = pytest.mark.gpu


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


def torus_desc(capi, n, a, inc_deg, **kw):
    img = capi.image_desc(n, n, a, inc_deg / 180.0 * math.pi)
    d = capi.TorusDesc(img=img, r0=kw.get("r0", 100.0), dl_max=kw.get("dl_max", 1e9),
                       precision=kw.get("precision", 1.0), options=kw.get("options", 0),
                       max_steps=kw.get("max_steps", 20000), max_error=kw.get("max_error", 1e-2), r_stop_in=kw.get("r_stop_in", 1.05), r_stop_out=1.01,
                       shape=kw.get("shape", 0), torus_r=kw.get("torus_r", 8.0), torus_w=kw.get("torus_w", 2.0),
                       torus_l=kw.get("torus_l", 3.5), emis0=kw.get("emis0", 1.0), absorb0=kw.get("absorb0", 0.0))
    return d


def run_torus(capi, d):
    n = d.img.nx * (d.img.y1 - d.img.y0)
    st = capi.DeviceBuffer(n * 40); steps = capi.DeviceBuffer(n * 4); xe = capi.DeviceBuffer(n * 32)
    ce = capi.DeviceBuffer(n * 8); me = capi.DeviceBuffer(n * 4)
    capi.torus_image_device(d, st.ptr, aux={"steps": steps.ptr, "x_end": xe.ptr, "carter_error": ce.ptr,
                                            "max_step_error": me.ptr})
    capi.synchronize()
    return (st.to_numpy(np.float64, (n, 5)), steps.to_numpy(np.int32, (n,)), xe.to_numpy(np.float64, (n, 4)),
            ce.to_numpy(np.float64, (n,)), me.to_numpy(np.float32, (n,)))


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


def test_torus_kernel_matches_cpu_integration(capi):
    """Kerr, a=0.9: final position and step count per ray against the oracle's own step loop;
    optically thin torus intensity against the same accumulation done on the CPU trace."""
    n, a, inc, r0 = 24, 0.9, 70.0, 100.0
    d = torus_desc(capi, n, a, inc, r0=r0)
    S, steps, xe, ce, me = run_torus(capi, d)
    orc = ol.Oracle()
    rmax = orc.r_ms(a) + 8.0
    c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * rmax
    cases = [(a, inc / 180.0 * math.pi, c[ix], c[iy], r0, 1.0, 0) for iy in range(n) for ix in range(n)]
    res = gga.verlet_traces(ol.ORACLE_SO, "orc_", cases, 20000)
    same_steps = 0
    for i, (m, tr, xs, ks, car) in enumerate(res):
        if m <= 0:
            assert steps[i] == 0
            continue
        same_steps += int(m == steps[i])
        if m == steps[i]:
            assert abs(xe[i, 1] - tr[m - 1, 1]) <= 1e-6 * max(1.0, abs(tr[m - 1, 1])), (i, xe[i], tr[m - 1, :4])
    # rounding differences in cos/acos may shift a float threshold for a few rays; the bulk must agree
    assert same_steps >= 0.97 * len(res), same_steps
    assert np.isfinite(S[:, 0]).all() and (S[:, 0] >= 0).all() and S[:, 0].max() > 0


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
        d = capi.image_desc(n, n, a, math.radians(inc), pol_degree=0.1, strict=strict)
        N = n * n
        st = capi.DeviceBuffer(3 * N * 8); chi = capi.DeviceBuffer(N * 8); gg = capi.DeviceBuffer(N * 8)
        capi.disk_image_polarized_device(d, st.ptr, chi.ptr, aux={"g": gg.ptr})
        capi.synchronize()
        CH = chi.to_numpy(np.float64, (N,)); G = gg.to_numpy(np.float64, (N,))
        rmax = ol.Oracle().r_ms(a) + 8.0
        c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * rmax
        al = np.ascontiguousarray(np.tile(c, n)); be = np.ascontiguousarray(np.repeat(c, n))
        rchi = np.zeros(N); rr = np.zeros(N); rg = np.zeros(N); rwp = np.zeros((N, 2))
        rc = drv.cpu_polarized_rays(ol.ORACLE_SO.encode(), b"orc_", a, math.radians(inc), -1.0, N, al.ctypes.data, be.ctypes.data,
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
        assert same >= (0.99 if strict else 0.95) * len(res), (a, inc, prec, strict, same, len(res))
        assert np.isfinite(S[:, 0]).all()
