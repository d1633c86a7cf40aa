"""The CPU restatement (oracle/) against the golden vectors captured from the unmodified
reference (oracle/gen_golden.py).  This is what pins the oracle; it runs without a GPU.

Both sides were produced by the same compiler family on the same libm without FMA contraction,
so agreement is expected to the last bit; the assertions allow 4 ulp so that a different glibc
on another box does not turn a libm difference into a failure, and report exact-match counts.
"""
import ctypes as C
import math

import numpy as np
import pytest

import oraclelib as ol

ULP4 = 4 * np.finfo(np.float64).eps


def close(a, b, rtol=ULP4, what=""):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, what
    both_nan = np.isnan(a) & np.isnan(b)
    both_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    ok = both_nan | both_inf | (np.abs(a - b) <= rtol * np.maximum(np.abs(a), np.abs(b))) | (a == b)
    assert ok.all(), "%s: %d/%d differ, worst rel %.3e" % (
        what, (~ok).sum(), ok.size,
        np.nanmax(np.abs(a - b)[~ok] / np.maximum(np.abs(b[~ok]), 1e-300)))


def test_carlson_and_jacobi(oracle, golden):
    g = golden("kat_elliptic.npz")
    close([oracle.rf(*v) for v in zip(g["c_x"], g["c_y"], g["c_z"])], g["rf"], what="rf")
    close([oracle.rd(*v) for v in zip(g["c_x"], g["c_y"], g["rd_z"])], g["rd"], what="rd")
    close([oracle.rc(*v) for v in zip(g["rc_x"], g["rc_y"])], g["rc"], what="rc")
    close([oracle.rj(*v) for v in zip(g["rj_x"], g["rj_y"], g["rj_z"], g["c_p"])], g["rj"], what="rj")
    close([oracle.elliptic_k(v) for v in g["k_m"]], g["elliptic_k"], what="elliptic_k")
    close([oracle.jacobi_isn(*v) for v in zip(g["isn_z"], g["k_m"])], g["jacobi_isn"], what="isn")
    close([oracle.jacobi_icn(*v) for v in zip(g["icn_z"], g["k_m"])], g["jacobi_icn"], what="icn")
    close([oracle.jacobi_itn(*v) for v in zip(g["itn_z"], g["k_m"])], g["jacobi_itn"], what="itn")
    s, c, d = C.c_double(), C.c_double(), C.c_double()
    sn, cn, dn = [], [], []
    for u, m in zip(g["sn_u"], g["k_m"]):
        oracle.jacobi_sncndn(u, m, C.byref(s), C.byref(c), C.byref(d))
        sn.append(s.value); cn.append(c.value); dn.append(d.value)
    close(sn, g["sn"], what="sn"); close(cn, g["cn"], what="cn"); close(dn, g["dn"], what="dn")


GEOD_F64 = ["a", "alpha", "beta", "incl", "cos_i", "l", "q", "m2p", "m2m", "mm", "mK", "rp", "Rpc", "Tpp", "Tip"]


def test_geodesic_records(oracle, golden):
    g = golden("kat_geodesic.npz")
    n = len(g["inp"])
    exact = 0
    for i in range(n):
        inc, a, al, be = g["inp"][i]
        gd = ol.Geodesic(); C.memset(C.byref(gd), 0, 240)
        e = C.c_int(-1)
        ok = oracle.geodesic_init_inf(inc, a, al, be, C.byref(gd), C.byref(e))
        assert ok == g["ok"][i] and e.value == g["err"][i], (i, ok, e.value)
        ref = ol.Geodesic.from_buffer_copy(g["dump"][i].tobytes())
        if ok:
            assert (gd.nrr, gd.type) == (ref.nrr, ref.type), i
            for f in GEOD_F64:
                close(getattr(gd, f), getattr(ref, f), what="geodesic.%s[%d]" % (f, i))
            for f in ("r1", "r2", "r3", "r4"):
                close([getattr(gd, f).re, getattr(gd, f).im], [getattr(ref, f).re, getattr(ref, f).im], what=f)
            exact += ol.struct_bytes(gd)[:200] == ol.struct_bytes(ref)[:200]
            close(oracle.geodesic_find_midplane_crossing(C.byref(gd), 0), g["P0"][i], what="P0")
            close(oracle.geodesic_find_midplane_crossing(C.byref(gd), 1), g["P1"][i], what="P1")
            if not math.isnan(g["P0"][i]):
                close(oracle.geodesic_position_rad(C.byref(gd), g["P0"][i]), g["r0"][i], what="r0")
            if not math.isnan(g["P1"][i]):
                close(oracle.geodesic_position_rad(C.byref(gd), g["P1"][i]), g["r1"][i], what="r1")
            if not math.isnan(g["Pq0"][i]):
                close(oracle.geodesic_P_int(C.byref(gd), g["rq"][i], 0), g["Pq0"][i], what="P_int0")
            if not math.isnan(g["Pq1"][i]):
                close(oracle.geodesic_P_int(C.byref(gd), g["rq"][i], 1), g["Pq1"][i], what="P_int1")
            if not math.isnan(g["Pm"][i]):
                close(oracle.geodesic_position_pol(C.byref(gd), g["Pm"][i]), g["mpol"][i], what="pol")
                close(oracle.geodesic_dm_sign(C.byref(gd), g["Pm"][i]), g["dms"][i], what="dm_sign")
                k = ol.D4()
                oracle.geodesic_momentum(C.byref(gd), g["Pm"][i], g["rmom"][i], g["mpol"][i], k)
                close(list(k), g["kmom"][i], what="momentum")
    nok = int(g["ok"].sum())
    assert exact == nok, "only %d of %d records are byte-identical with the reference" % (exact, nok)


def test_geodesic_init_src_records(oracle, golden):
    """geodesic_init_src (ref src/sim5kerr-geod.c:106-173): 2 000 records from the reference -- the round trip
    init_inf -> point of the trajectory -> momentum -> init_src of ref src/sim5unittests.c:171-255 (both sides of the
    pericentre, spin 0 for the 1e-8 clamp), random ZAMO-frame photons (RR_BH, CC, captured rays) and the error
    returns (q = 0, mu_plus and mu_0 out of range).  Byte level: return value, error code and every field the
    reference writes (the first 200 bytes of the record; the rest is never written)."""
    g = golden("kat_init_src.npz")
    inp = g["inp"]
    exact = 0
    for i in range(len(inp)):
        gd = ol.Geodesic(); C.memset(C.byref(gd), 0, 240)
        e = C.c_int(-1)
        ok = oracle.geodesic_init_src(inp[i, 0], inp[i, 1], inp[i, 2], ol.D4(*inp[i, 3:7]), int(inp[i, 7]),
                                      C.byref(gd), C.byref(e))
        assert ok == g["ok"][i] and e.value == g["err"][i], (i, ok, e.value, g["ok"][i], g["err"][i])
        same = ol.struct_bytes(gd)[:200] == g["dump"][i].tobytes()[:200]
        if not same:
            ref = ol.Geodesic.from_buffer_copy(g["dump"][i].tobytes())
            assert (gd.nrr, gd.type) == (ref.nrr, ref.type), i
            for f in GEOD_F64:
                close(getattr(gd, f), getattr(ref, f), what="init_src.%s[%d]" % (f, i))
        exact += same
    assert exact == len(inp), "only %d of %d init_src records are byte-identical with the reference" % (exact, len(inp))
    # the round trip gives the observer back (the reference's own acceptance test, ref src/sim5unittests.c:239)
    rt = (inp[:, 8] == 0) & (inp[:, 0] > 1e-3)
    rec = np.frombuffer(g["dump"].tobytes(), dtype=np.dtype([("f", np.float64, 30)]))["f"]
    assert np.max(np.abs(rec[rt, 4] - np.cos(inp[rt, 9]))) < 1e-5
    assert np.max(np.abs(rec[rt, 1] - inp[rt, 10])) < 1e-4 and np.max(np.abs(rec[rt, 2] - inp[rt, 11])) < 1e-4


AZM_FUNCS = ["elliptic_f_cos", "elliptic_e_cos", "elliptic_pi_complete", "elliptic_pi_cos", "integral_C2",
             "integral_C2_cos", "integral_Z1", "integral_Z2", "integral_Rm1", "integral_Rm2", "integral_R1",
             "integral_R2", "integral_R_r0_re", "integral_R_r0_re_inf", "integral_R_r1_re", "integral_R_r2_re",
             "integral_R_rp_re", "integral_R_rp_re_inf", "integral_R_r0_cc", "integral_R_r0_cc_inf",
             "integral_R_r1_cc", "integral_R_r2_cc", "integral_R_rp_cc2", "integral_R_rp_cc2_inf",
             "integral_T_m0", "integral_T_m2", "integral_T_mp"]


def test_azimuth_and_timedelay(oracle, golden):
    """geodesic_position_azm, geodesic_timedelay and the 27 integrals under them."""
    g = golden("kat_azimuth.npz")
    for name in AZM_FUNCS:
        fn = getattr(oracle, name)
        close([fn(*row) for row in g["in_" + name]], g["out_" + name], what=name)
    n = len(g["inp"])
    phi = np.full(n, np.nan); dt_a = np.full(n, np.nan); dt_e = np.full(n, np.nan)
    for i in range(n):
        if g["gtype"][i] not in (40, 2):
            continue
        inc, a, al, be = g["inp"][i]
        gd = ol.Geodesic(); e = C.c_int(-1)
        assert oracle.geodesic_init_inf(inc, a, al, be, C.byref(gd), C.byref(e))
        if not (math.isnan(g["r1"][i]) or math.isnan(g["m1"][i])):
            phi[i] = oracle.geodesic_position_azm(C.byref(gd), g["r1"][i], g["m1"][i], g["P1"][i])
        dt_a[i] = oracle.geodesic_timedelay(C.byref(gd), g["P1"][i], 0.0, 0.0, g["P2"][i], 0.0, 0.0)
        if not (math.isnan(g["r1"][i]) or math.isnan(g["r2"][i])):
            dt_e[i] = oracle.geodesic_timedelay(C.byref(gd), g["P1"][i], g["r1"][i], g["m1"][i],
                                                g["P2"][i], g["r2"][i], g["m2"][i])
    close(phi, g["phi"], what="position_azm")
    close(dt_a, g["dt_auto"], what="timedelay (r, m from P)")
    close(dt_e, g["dt_expl"], what="timedelay (explicit r, m)")
    assert np.isfinite(g["phi"]).sum() > 2000 and np.isfinite(g["dt_auto"]).sum() > 2000


def test_vectors(oracle, golden):
    """dotprod (Kerr / flat), vector_norm_to (time-like, null, space-like targets; NaN where the reference gives NaN) and
    Omega_from_ell of the CPU restatement against the reference's values, 800 random metrics"""
    g = golden("kat_vectors.npz")
    for i in range(len(g["a"])):
        mt = ol.Metric()
        oracle.kerr_metric(g["a"][i], g["r"][i], g["m"][i], C.byref(mt))
        close(np.frombuffer(ol.struct_bytes(mt), np.float64), g["metric"][i], what="metric")
        v1, v2 = ol.D4(*g["v1"][i]), ol.D4(*g["v2"][i])
        close(oracle.dotprod(v1, v2, C.byref(mt)), g["dot"][i], what="dotprod")
        close(oracle.dotprod(v1, v2, None), g["dot_flat"][i], what="dotprod flat")
        w = ol.D4(*g["v1"][i]); oracle.vector_norm_to(w, g["norm"][i], C.byref(mt))
        assert np.array_equal(np.isnan(list(w)), np.isnan(g["vn"][i]))
        if not np.isnan(g["vn"][i]).any():
            close(list(w), g["vn"][i], what="vector_norm_to")
        w = ol.D4(*g["v1"][i]); oracle.vector_norm_to(w, g["norm"][i], None)
        close(list(w), g["vn_flat"][i], what="vector_norm_to flat")
        close(oracle.Omega_from_ell(g["ell"][i], C.byref(mt)), g["Omega"][i], what="Omega_from_ell")


def test_kerr(oracle, golden):
    g = golden("kat_kerr.npz")
    n = len(g["a"])
    for i in range(n):
        a, r, m = g["a"][i], g["r"][i], g["m"][i]
        mt = ol.Metric(); mc = ol.Metric(); t = ol.Tetrad(); G = ol.G444()
        oracle.kerr_metric(a, r, m, C.byref(mt))
        close(np.frombuffer(ol.struct_bytes(mt), np.float64), g["metric"][i], what="metric")
        oracle.kerr_metric_contravariant(a, r, m, C.byref(mc))
        close(np.frombuffer(ol.struct_bytes(mc), np.float64), g["metric_contra"][i], what="metric_contra")
        oracle.kerr_connection(a, r, m, G)
        close(np.frombuffer(bytes(memoryview(G)), np.float64), g["connection"][i], what="connection")
        oracle.tetrad_zamo(C.byref(mt), C.byref(t))
        close(np.frombuffer(ol.struct_bytes(t), np.float64), g["zamo"][i], what="zamo")
        oracle.tetrad_azimuthal(C.byref(mt), g["Omega"][i], C.byref(t))
        close(np.frombuffer(ol.struct_bytes(t), np.float64), g["azim"][i], what="azimuthal")
        oracle.tetrad_surface(C.byref(mt), g["Omega"][i], g["V"][i], g["dhdr"][i], C.byref(t))
        close(np.frombuffer(ol.struct_bytes(t), np.float64), g["surf"][i], what="surface")
        vi = ol.D4(*g["vin"][i]); vo = ol.D4()
        oracle.bl2on(vi, vo, C.byref(t)); close(list(vo), g["v_on"][i], what="bl2on")
        oracle.on2bl(vi, vo, C.byref(t)); close(list(vo), g["v_bl"][i], what="on2bl")
        k = ol.D4()
        oracle.photon_momentum(a, r, m, g["l"][i], g["q"][i], g["r_sign"][i], g["m_sign"][i], k)
        close(list(k), g["kph"][i], what="photon_momentum")
        if not math.isnan(g["kph"][i, 0]):
            L, Q = C.c_double(), C.c_double()
            oracle.photon_motion_constants(a, r, m, k, C.byref(L), C.byref(Q))
            close([L.value, Q.value], [g["L"][i], g["Q"][i]], what="motion constants")
            close(oracle.photon_carter_const(k, C.byref(mt)), g["Qcarter"][i], what="carter")
        close(oracle.gfactorK(g["gK_r"][i], a, g["gK_l"][i]), g["gK"][i], what="gfactorK")
        close(oracle.OmegaK(r, a), g["OmegaK"][i], what="OmegaK")
        close(oracle.ellK(r, a), g["ellK"][i], what="ellK")
    close([oracle.r_ms(a) for a in g["r_ms_a"]], g["r_ms"], rtol=0, what="r_ms")
    close([oracle.r_bh(a) for a in g["r_ms_a"]], g["r_bh"], rtol=0, what="r_bh")


def test_disk_nt(oracle, golden):
    g = golden("kat_disk.npz")
    for j, a in enumerate(g["spins"]):
        oracle.disk_nt_setup(10.0, a, 0.1, 0.1)
        close(oracle.disk_nt_r_min(), g["rmin_%d" % j][0], rtol=0, what="r_min")
        close([oracle.disk_nt_flux(r) for r in g["r_%d" % j]], g["flux_%d" % j], what="flux a=%g" % a)
        close([oracle.disk_nt_ell(r) for r in g["r_%d" % j]], g["ell_%d" % j], what="ell")
        # the band rms <= r <= rms_disk carries zero flux (float-rounded inner edge)
        assert (g["flux_%d" % j][g["r_%d" % j] <= g["rmin_%d" % j][0] - 2e-3] == 0).all()
    oracle.disk_nt_setup(3.7e6, 0.7, 0.31, 0.05)
    close([oracle.disk_nt_flux(r) for r in g["r_x"]], g["flux_x"], what="flux other M, mdot")


def test_disk_flux_inner_edge_band_bit_for_bit(oracle, golden):
    """the band where the closed form of the flux is the reference's rounding pattern (negative values and exact zeros
    included; oracle/gen_golden.py:kat_disk_edge): the restatement returns the reference's BITS on all 13 x 2 065 radii -- and,
    in the build container, on 100 000 more per spin against the live library"""
    g = golden("kat_disk_edge.npz")
    for j, (M, a, mdot, al) in enumerate(g["models"]):
        got = ol.cpu_disk_flux(g["r_%d" % j], float(a), kind="port", M=float(M), mdot=float(mdot), alpha_visc=float(al))
        assert np.array_equal(got.view(np.uint64), g["flux_%d" % j].view(np.uint64)), j
    if ol.have_reference():
        rng = np.random.default_rng(67)
        for a in (0.0, 0.3, 0.9, 0.998, 0.9999):
            ref = ol.Reference(); ref.disk_nt_setup(10.0, a, 0.1, 0.1, 0)
            edge = float(np.float32(ref.disk_nt_r_min()))
            r = edge + 10.0 ** rng.uniform(-14, -2, 100000)
            assert np.array_equal(ol.cpu_disk_flux(r, a, kind="port").view(np.uint64), ol.cpu_disk_flux(r, a, kind="reference").view(np.uint64)), a


def test_polarization_and_blackbody(oracle, golden):
    g = golden("kat_polar.npz")
    for i in range(len(g["a"])):
        mt = ol.Metric.from_buffer_copy(g["metric"][i].tobytes())
        k = ol.D4(*g["k"][i]); f = ol.D4(*g["f"][i])
        w = oracle.polarization_constant(k, f, C.byref(mt))
        close([w.re, w.im], g["wp"][i], what="polarization_constant")
        fo = ol.D4()
        oracle.polarization_vector(k, ol.Cplx(*g["wp"][i]), C.byref(mt), fo)
        close(list(fo), g["f_back"][i], what="polarization_vector")
        w = oracle.polarization_constant_infinity(g["a"][i], g["alpha"][i], g["beta"][i], g["incl"][i])
        close([w.re, w.im], g["wp_inf"][i], what="constant_infinity")
        close(oracle.polarization_angle_rotation(g["a"][i], g["incl"][i], g["alpha"][i], g["beta"][i],
                                                 ol.Cplx(*g["wp"][i])), g["rot"][i], what="angle_rotation")
        close(oracle.blackbody_Iv(g["T"][i], g["hardf"][i], g["cos_mu"][i], g["E"][i]), g["Iv"][i], what="Iv")


def test_raytrace_api(oracle, golden):
    g = golden("kat_raytrace_api.npz")
    for i in range(len(g["a"])):
        x = ol.D4(*g["x"][i]); k = ol.D4(*g["k"][i])
        rtd = ol.RaytraceData(); C.memset(C.byref(rtd), 0, 144)
        oracle.raytrace_prepare(g["a"][i], x, k, g["precision"][i], int(g["options"][i]), C.byref(rtd))
        ref0 = ol.RaytraceData.from_buffer_copy(g["rtd_prepared"][i].tobytes())
        assert (rtd.opt_gr, rtd.pass_) == (ref0.opt_gr, ref0.pass_)
        close([rtd.step_epsilon, rtd.E, rtd.Q, rtd.kt] + list(rtd.dk),
              [ref0.step_epsilon, ref0.E, ref0.Q, ref0.kt] + list(ref0.dk), what="prepare")
        s = C.c_double(g["stepcap"][i])
        oracle.raytrace(x, k, C.byref(s), C.byref(rtd))
        ref1 = ol.RaytraceData.from_buffer_copy(g["rtd_stepped"][i].tobytes())
        close(list(x), g["x1"][i], what="x after step"); close(list(k), g["k1"][i], what="k after step")
        close(s.value, g["step"][i], what="step taken")
        assert rtd.pass_ == ref1.pass_
        close([rtd.kt, rtd.error] + list(rtd.dk), [ref1.kt, ref1.error] + list(ref1.dk), what="rtd after step")
        close(oracle.raytrace_error(x, k, C.byref(rtd)), g["carter"][i], rtol=1e-9, what="carter error")


def test_raytrace_sequences(oracle, golden):
    """Whole step sequences: same number of steps, same states at the sampled steps."""
    import gen_golden_access as gga
    g = golden("kat_raytrace.npz")
    cases = g["cases"]
    res = gga.verlet_traces(ol.ORACLE_SO, "orc_", [tuple(c[:6]) + (int(c[6]),) for c in cases], 6000)
    for i, (n, tr, xs, ks, car) in enumerate(res):
        assert n == int(g["n_%d" % i][0]), "ray %d: %d steps, reference %d" % (i, n, g["n_%d" % i][0])
        close(xs, g["x0_%d" % i], what="x start"); close(ks, g["k0_%d" % i], what="k start")
        close(tr[g["idx_%d" % i]], g["tr_%d" % i], what="trace %d" % i)
        close(car, g["carter_%d" % i][0], rtol=1e-9, what="carter")


IMAGES = [("img_c2_1024_a0998_i70.npz", 8), ("img_c3_2048_a09_i70.npz", 16)]


def test_image_c1_complete(golden):
    g = golden("img_c1_64_a0_i60.npz")
    o = ol.cpu_disk_image("port", 64, 64, 0.0, 60.0, nthreads=1, full=True)
    for k in ("cls", "gtype", "image_f", "image_g"):
        assert np.array_equal(o[k], g[k]), k
    for k in ("r", "g", "flux"):
        close(o[k], g[k], what=k)
    # the known answers of BASELINE.md for this configuration
    assert np.bincount(g["cls"].ravel(), minlength=6).tolist() == [0, 0, 3510, 184, 34, 368]


@pytest.mark.parametrize("name,threads", IMAGES)
def test_image_classes_and_samples(golden, name, threads):
    g = golden(name)
    n, a, inc, dec = int(g["n"][0]), float(g["a"][0]), float(g["inc_deg"][0]), int(g["dec"][0])
    o = ol.cpu_disk_image("port", n, n, a, inc, nthreads=threads, full=True)
    assert np.array_equal(o["cls"], g["cls"]), "class map differs in %d pixels" % (o["cls"] != g["cls"]).sum()
    sl = (slice(dec // 2, None, dec), slice(dec // 2, None, dec))
    close(o["r"][sl], g["d_r"], what="r"); close(o["g"][sl], g["d_g"], what="g"); close(o["flux"][sl], g["d_flux"], what="F")
    assert np.array_equal(o["image_f"][sl], g["d_image_f"]) and np.array_equal(o["image_g"][sl], g["d_image_g"])
    close(o["g"].sum(dtype=np.float64), g["sum_g"][0], rtol=1e-12, what="sum g")
    close((o["flux"] * o["g"] ** 4).sum(dtype=np.float64), g["sum_fg4"][0], rtol=1e-12, what="sum F g^4")


def test_image_headline_rows(golden):
    """4096^2 headline image: a band of rows through the shadow (the full map is the GPU test's job)."""
    g = golden("img_head_4096_a0998_i70.npz")
    y0, y1 = 2016, 2080
    o = ol.cpu_disk_image("port", 4096, 4096, 0.998, 70.0, y0=y0, y1=y1, nthreads=8, full=False)
    assert np.array_equal(o["cls"], g["cls"][y0:y1])
    assert g["counts"].tolist() == [0, 74725, 15865362, 371216, 0, 465913]
    assert g["type_counts"].tolist() == [13054020, 3713076, 10120, 0]


def test_polarized_recipe(golden):
    import gen_golden_access as gga
    g = golden("img_c3_polarized.npz")
    chi, r, gg, wp = gga.polarized_rays(ol.ORACLE_SO, "orc_", float(g["a"][0]), float(g["inc_deg"][0]),
                                        g["alpha"], g["beta"])
    assert np.array_equal(np.isnan(chi), np.isnan(g["chi"]))
    close(chi, g["chi"], rtol=1e-13, what="chi"); close(r, g["r"], what="r"); close(gg, g["g"], what="g")
    close(wp, g["wp"], rtol=1e-12, what="kappa")


def test_torus_c4_port_equals_reference(golden):
    """C4 subset (every 16th pixel of the 1024^2 grid, and the view as a 16 x 16 image at precision 0.01): our restatement's
    raytrace() loop with the per-step transfer (oracle/cpu_driver.c:cpu_torus_rays) against the same loop over the
    unmodified reference (golden torus_c4.npz): step counts, end states and transfer integrals bit for bit."""
    import gen_golden_access as gga
    g = golden("torus_c4.npz")
    n, a, inc = int(g["n"][0]), float(g["a"][0]), math.radians(float(g["inc_deg"][0]))
    rmax = ol.Oracle().r_ms(a) + 8.0
    c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * rmax
    for tag, kw in (("thin", {}), ("absorb", {"absorb0": 0.3}), ("fine", {"precision": 0.01, "max_steps": 50000})):
        src = "thin" if tag == "absorb" else tag
        sel = slice(None, None, 4) if tag != "fine" else slice(None)          # a quarter of the 4 096 rays is enough here
        grid = c if tag != "fine" else ((np.arange(16) + .5) / 16 - 0.5) * 2.0 * rmax       # "fine": a 16 x 16 image
        o = gga.torus_rays(ol.ORACLE_SO, "orc_", a, inc, grid[g[src + "_ix"]][sel], grid[g[src + "_iy"]][sel], **kw)
        for k in ("steps", "x_end", "k_end", "I", "tau", "carter", "max_step_error"):
            key = "%s_%s" % (tag if k in ("I", "tau") else src, k)
            assert np.array_equal(o[k], g[key][sel], equal_nan=True), (tag, k)
    assert g["absorb_tau"].max() > 5 and g["thin_I"].max() > 20


def test_torus_conditioning_probe(golden):
    """The conditioning probe of the GPU parity tests (cpu_torus_rays_perturbed): no shift = the plain loop bit for bit,
    restatement and reference move alike under the same +-1 ulp shift of the start state, and on the C4 sample the
    reference's own end states move by up to ~1e-8 for one unit in the last place -- the amplification of rays that
    wind around the photon orbit, which any two implementations see."""
    import gen_golden_access as gga
    g = golden("torus_c4.npz")
    n, a, inc = int(g["n"][0]), float(g["a"][0]), math.radians(float(g["inc_deg"][0]))
    rmax = ol.Oracle().r_ms(a) + 8.0
    c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * rmax
    order = np.arange(2, g["thin_steps"].size, 8)                      # 512 rays of the sample
    al, be = c[g["thin_ix"]][order], c[g["thin_iy"]][order]
    base = gga.torus_rays(ol.ORACLE_SO, "orc_", a, inc, al, be)
    zero = gga.torus_rays(ol.ORACLE_SO, "orc_", a, inc, al, be, ulps=[0] * 8)
    for k in base:
        assert np.array_equal(base[k], zero[k], equal_nan=True) and np.array_equal(base[k][:, ...], g["thin_" + k][order], equal_nan=True), k
    u = [0, 0, 0, 0, 0, 1, 0, 0]                                       # k^r one unit in the last place up
    port = gga.torus_rays(ol.ORACLE_SO, "orc_", a, inc, al, be, ulps=u)
    if ol.have_reference():
        refp = gga.torus_rays(ol.REF_SO, "", a, inc, al, be, ulps=u)
        for k in port:
            assert np.array_equal(port[k], refp[k], equal_nan=True), k
    assert np.array_equal(port["steps"], base["steps"])
    move = np.abs(port["x_end"][:, 1] / base["x_end"][:, 1] - 1)
    assert 1e-10 < move.max() < 1e-6 and (move > 0).mean() > 0.5, (move.max(), (move > 0).mean())   # worst ray: ~1e7 x the shift


def test_disk_model_rest(oracle, golden):
    """disk_nt_mdot / disk_nt_lumi / disk_nt_sigma incl. the luminosity-parametrised set-up (bisection over the Simpson
    integral of the flux), and r_ph / r_mb: bit for bit against the reference."""
    g = golden("kat_disk_model.npz")
    for j, (M, a, x, al, opt) in enumerate(g["setups"]):
        oracle.disk_nt_setup(M, a, x, al, int(opt))
        assert oracle.disk_nt_mdot() == g["mdot_%d" % j][0], j
        assert oracle.disk_nt_lumi() == g["lumi_%d" % j][0], j
        close(oracle.disk_nt_r_min(), g["rmin_%d" % j][0], rtol=0, what="r_min")
        close([oracle.disk_nt_sigma(r) for r in g["r_%d" % j]], g["sigma_%d" % j], rtol=0, what="sigma %d" % j)
        close([oracle.disk_nt_flux(r) for r in g["r_%d" % j]], g["flux_%d" % j], rtol=0, what="flux %d" % j)
        assert g["sigma_%d" % j][0] == 0 and g["sigma_%d" % j][2] > 0
    close([oracle.r_ph(a) for a in g["spin"]], g["r_ph"], rtol=0, what="r_ph")
    close([oracle.r_mb(a) for a in g["spin"]], g["r_mb"], rtol=0, what="r_mb")
    # luminosity set-ups give back the luminosity asked for (the reference's bisection stops at 1e-6 in mdot)
    for j, (M, a, x, al, opt) in enumerate(g["setups"]):
        if opt:
            assert abs(g["lumi_%d" % j][0] / x - 1) < 1e-4


def test_kerr_newman_restatement_is_bit_identical_to_the_reference(oracle, golden):
    """kerr_newman_metric / _contravariant / _connection (ref src/sim5kerr.c:136-194, 321-397): the restatement gives the
    unmodified reference's bytes on all 500 golden points (oracle/gen_golden.py:kat_kerr_newman)"""
    g = golden("kat_kerr_newman.npz")
    for i in range(len(g["a"])):
        a, Q, r, m = (float(g[k][i]) for k in ("a", "Q", "r", "m"))
        mt = ol.Metric(); G = ol.G444()
        oracle.kerr_newman_metric(a, Q, r, m, C.byref(mt))
        assert np.array_equal(np.frombuffer(ol.struct_bytes(mt), np.float64), g["metric"][i]), i
        oracle.kerr_newman_metric_contravariant(a, Q, r, m, C.byref(mt))
        assert np.array_equal(np.frombuffer(ol.struct_bytes(mt), np.float64), g["metric_contra"][i]), i
        oracle.kerr_newman_connection(a, Q, r, m, G)
        assert np.array_equal(np.frombuffer(bytes(memoryview(G)), np.float64), g["connection"][i]), i
