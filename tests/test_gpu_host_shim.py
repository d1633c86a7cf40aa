"""The SIM5 scalar API served by the GPU library (n = 1 batch calls), driven from C."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

import oraclelib as ol
from gpuutil import assert_close, assert_flux, deg2rad

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "sim5_amd", "host")


def test_scalar_api_program_matches_oracle(tmp_path, capi):
    exe = str(tmp_path / "probe")
    subprocess.run(["gcc", os.path.join(ROOT, "tests", "c", "shim_probe.c"), os.path.join(HOST, "sim5lib.c"),
                    "-I", HOST, "-o", exe, "-lm", "-O3", "-w", "-fgnu89-inline"], check=True)
    n, a, inc = 20, 0.9, 65.0
    env = dict(os.environ, SIM5GPU_LIB=capi.LIB_PATH)
    p = subprocess.run([exe, str(a), str(inc), str(n)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    head = lines[0].split()
    orc = ol.Oracle()
    orc.disk_nt_setup(10.0, a, 0.1, 0.1)
    # r_ms goes through the device cbrt (1 ulp from glibc's); r_min is folded on the host like the reference
    assert abs(float(head[2]) / orc.r_ms(a) - 1) < 1e-15 and float(head[4]) == orc.disk_nt_r_min()
    assert abs(float(head[6]) / orc.r_bh(a) - 1) < 1e-15
    rec = np.array([[float(v) for v in ln.split()] for ln in lines[1:1 + n * n]])
    c = ol.cpu_disk_image("port", n, n, a, inc, nthreads=2, full=True)
    hit = np.where(c["cls"] == 2, 1, np.where(c["cls"] == 4, 2, 0)).ravel()
    assert np.array_equal(rec[:, 3].astype(int), hit)
    m = hit > 0
    assert_close(rec[m, 4], c["r"].ravel()[m], what="r"); assert_close(rec[m, 5], c["g"].ravel()[m], what="g")
    assert_flux(rec[m, 6], rec[m, 4], c["flux"].ravel()[m], a, what="flux")
    tail = lines[-3].split()
    assert tail[1] == "verlet" and int(tail[3]) > 100 and float(tail[7]) < 1e-2
    # the integrals of sim5elliptic.h and vector_norm_to through the scalar API
    tail = lines[-1].split()
    assert tail[1] == "ints"
    want = [orc.elliptic_f_sin(0.7, 0.4), orc.elliptic_pi_cos(0.35, -1.7, 0.62),
            orc.integral_R_rp_cc2(4.0, 1.5, 0.8, 1.1, 1.2, 4.5, 30.0), orc.integral_T_mp(3.0, 0.7, 1.0, -0.4), 1.0]
    for got, ref in zip(tail[2:7], want):
        assert abs(float(got) / ref - 1) < 1e-9, (got, ref)
    # geodesic_position_azm / geodesic_timedelay through the scalar API against the CPU checker
    tail = lines[-2].split()
    gd = ol.Geodesic(); e = C.c_int(0)
    assert orc.geodesic_init_inf(deg2rad(inc), a, 6.0, 5.0, C.byref(gd), C.byref(e))
    P1, P2 = 0.6 * gd.Rpc, 1.4 * gd.Rpc
    r1 = orc.geodesic_position_rad(C.byref(gd), P1); m1 = orc.geodesic_position_pol(C.byref(gd), P1)
    assert abs(float(tail[2]) / orc.geodesic_position_azm(C.byref(gd), r1, m1, P1) - 1) < 1e-9
    assert abs(float(tail[4]) / orc.geodesic_timedelay(C.byref(gd), P1, 0.0, 0.0, P2, 0.0, 0.0) - 1) < 1e-9


def _cc(tmp_path, src, name, extra=()):
    exe = str(tmp_path / name)
    subprocess.run(["gcc", os.path.join(ROOT, "tests", "c", src), os.path.join(HOST, "sim5lib.c"), "-I", HOST, "-o", exe, "-lm",
                    "-O3", "-w", "-fgnu89-inline"] + list(extra), check=True)
    return exe


def test_c1_through_the_scalar_api_one_round_trip_per_ray(tmp_path, capi, golden):
    """BASELINE.json configs[0] -- 64 x 64, a = 0, i = 60 deg, the loop of ref examples/04-disk-image-eqplane/disk-image.c:
    53-105 -- through the SIM5 SCALAR API on the GPU (tests/c/shim_probe.c is that loop, call for call), against the golden
    image of the unmodified reference (img_c1_64_a0_i60.npz: hit / miss exact, r, g, flux within 1e-6).  Run three ways: with
    the shim's look-ahead (records of whole rows asked for in one batch call, every call answered after a bit-for-bit check of
    its arguments: the default), with the per-ray record alone (SIM5_SHIM_NO_LOOKAHEAD=1: geodesic_init_inf brings the
    crossings, radii, g-factors and fluxes of its ray in the same launch) and call by call (SIM5_SHIM_NO_CHAIN=1: five round
    trips per ray).  All records are made by the strict routines the single calls run, so the three outputs must be the same
    TEXT, digit for digit.  The rates are printed."""
    import time
    exe = _cc(tmp_path, "shim_probe.c", "probe")
    n, a, inc = 64, 0.0, 60.0
    outs, secs = [], []
    for extra in ({}, {"SIM5_SHIM_NO_LOOKAHEAD": "1"}, {"SIM5_SHIM_NO_CHAIN": "1"}):
        env = dict(os.environ, SIM5GPU_LIB=capi.LIB_PATH, **extra)
        t0 = time.time()
        p = subprocess.run([exe, str(a), str(inc), str(n)], env=env, capture_output=True, text=True, timeout=900)
        secs.append(time.time() - t0)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(p.stdout)
    assert outs[0] == outs[2], "the look-ahead run differs from the call-by-call run"
    assert outs[1] == outs[2], "the record-served run differs from the call-by-call run"
    print("scalar API, %d rays (process start-up included): %.3e rays/s with the look-ahead, %.3e with one round trip per ray, "
          "%.3e call by call" % (n * n, n * n / secs[0], n * n / secs[1], n * n / secs[2]))
    g = golden("img_c1_64_a0_i60.npz")
    rec = np.array([[float(v) for v in ln.split()] for ln in outs[0].strip().splitlines()[1:1 + n * n]])
    hit = np.where(g["cls"] == 2, 1, np.where(g["cls"] == 4, 2, 0)).ravel()
    assert np.array_equal(rec[:, 3].astype(int), hit), "hit / miss differs from the reference on %d pixels" % int((rec[:, 3].astype(int) != hit).sum())
    assert hit.astype(bool).sum() == 3544
    m = hit > 0
    assert_close(rec[m, 4], g["r"].ravel()[m], what="r"); assert_close(rec[m, 5], g["g"].ravel()[m], what="g")
    assert_flux(rec[m, 6], rec[m, 4], g["flux"].ravel()[m], 0.0, what="flux")
    # pixels the reference rejects (error from geodesic_init_inf) are rejected here with the same code
    err_ref = (g["cls"] == 0).ravel()
    assert np.array_equal(rec[:, 2] != 0, err_ref)


def test_lookahead_is_order_independent(tmp_path, capi):
    """The shim's look-ahead predicts the caller's NEXT pixels; it must never change what a call returns.  tests/c/shim_order.c
    asks for the pixels of an odd-sized, non-square image (37 x 23: central column and row included) in raster order, column by
    column, shuffled, raster with every third pixel skipped and caught up later, and as two images interleaved row by row: every
    order must print, pixel for pixel, the text of the raster run with the look-ahead switched off."""
    exe = _cc(tmp_path, "shim_order.c", "order")
    a, inc, nx, ny = 0.9, 65.0, 37, 23
    env = dict(os.environ, SIM5GPU_LIB=capi.LIB_PATH)
    run = lambda spin, order, **kw: subprocess.run([exe, str(spin), str(inc), str(nx), str(ny), str(order)], env=dict(env, **kw),
                                                   capture_output=True, text=True, timeout=600)
    base = run(a, 0, SIM5_SHIM_NO_LOOKAHEAD="1")
    assert base.returncode == 0 and len(base.stdout.splitlines()) == nx * ny, base.stderr[-1000:]
    for order in (0, 1, 2, 3):
        p = run(a, order)
        assert p.returncode == 0, p.stderr[-1000:]
        assert p.stdout == base.stdout, "order %d: output differs from the raster run without look-ahead" % order
    half = run(a, 5, SIM5_SHIM_NO_LOOKAHEAD="1")
    p = run(a, 4)
    first = "\n".join(l for l in p.stdout.splitlines() if not l.startswith("second")) + "\n"
    second = "\n".join(l[len("second "):] for l in p.stdout.splitlines() if l.startswith("second")) + "\n"
    assert first == base.stdout and second == half.stdout, "interleaved images differ from the separate runs"
    # and a larger raster image, where several rows are asked for in one call: against the per-ray record
    big = [subprocess.run([exe, "0.998", "70", "160", "96", "0"], env=dict(env, **kw), capture_output=True, text=True, timeout=900)
           for kw in ({}, {"SIM5_SHIM_NO_LOOKAHEAD": "1"})]
    assert big[0].returncode == 0 and big[0].stdout == big[1].stdout and len(big[0].stdout.splitlines()) == 160 * 96


def test_scalar_api_from_eight_host_threads(tmp_path, capi):
    """ref README.md:16,202: the per-ray functions are callable concurrently from any host thread.  tests/c/shim_threads.c:
    eight threads share one 48 x 48 image (rows dealt round robin), each through the scalar API -- its own record, look-ahead,
    staging memory and stream -- and each makes two batch calls of the C-ABI between its rows, checked bit for bit against the
    scalar calls.  The text printed in image order must be the single-threaded run's."""
    libdir = os.path.dirname(capi.LIB_PATH)
    exe = _cc(tmp_path, "shim_threads.c", "threads", ["-I", os.path.join(ROOT, "include"), "-lpthread", "-L", libdir, "-lsim5gpu",
                                                     "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"])
    env = dict(os.environ, SIM5GPU_LIB=capi.LIB_PATH)
    outs = []
    for threads in (1, 8):
        p = subprocess.run([exe, "0.9", "65", "48", str(threads)], env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, (p.stdout[-300:], p.stderr[-1500:])
        outs.append(p.stdout)
    assert outs[0] == outs[1] and "0 mismatches" in outs[0] and len(outs[0].splitlines()) == 48 * 48 + 1


def test_boundary_prototypes_program(tmp_path, capi):
    """tests/c/boundary_probe.c: the public prototypes that are not on the inner path, called from C through
    sim5_amd/host/sim5lib.c (n = 1 batch calls on the GPU), every printed number against the UNMODIFIED reference
    (oracle/_ref/libsim5ref.so, compiled from the reference's sources in the build container; it travels to the GPU
    box) called with the same arguments."""
    if not ol.have_reference():
        pytest.skip("oracle/_ref/libsim5ref.so not present")
    exe = str(tmp_path / "bprobe")
    subprocess.run(["gcc", os.path.join(ROOT, "tests", "c", "boundary_probe.c"), os.path.join(HOST, "sim5lib.c"),
                    "-I", HOST, "-o", exe, "-lm", "-O3", "-w", "-fgnu89-inline"], check=True)
    env = dict(os.environ, SIM5GPU_LIB=capi.LIB_PATH)
    L = C.CDLL(ol.REF_SO)
    D, D4, PM, PT = ol.D, ol.D4, ol.PM, ol.PT

    def fn(name, res, *args):
        f = getattr(L, name); f.restype = res; f.argtypes = list(args); return f
    for (a, r, m) in ((0.9, 6.0, 0.3), (0.0, 9.0, -0.5), (0.998, 2.5, 0.0)):
        p = subprocess.run([exe, str(a), str(r), str(m)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        got = {ln.split()[0]: np.array([float(v) for v in ln.split()[1:]]) for ln in p.stdout.splitlines()}
        want = {}
        g, gc, gf, gfc, t = ol.Metric(), ol.Metric(), ol.Metric(), ol.Metric(), ol.Tetrad()
        fn("kerr_metric", None, D, D, D, PM)(a, r, m, C.byref(g))
        fn("kerr_metric_contravariant", None, D, D, D, PM)(a, r, m, C.byref(gc))
        fn("flat_metric", None, D, D, PM)(r, m, C.byref(gf))
        fn("flat_metric_contravariant", None, D, D, PM)(r, m, C.byref(gfc))
        mt = lambda x: np.frombuffer(ol.struct_bytes(x), np.float64)[3:8].copy()
        want["kerr_metric_contravariant"], want["flat_metric"], want["flat_metric_contravariant"] = mt(gc), mt(gf), mt(gfc)
        G = ol.G444(); k = D4(1.0, -0.6, 0.02, 0.01); dk = D4()
        fn("flat_connection", None, D, D, ol.G444)(r, m, G)
        null = fn("vector_norm_to_null", None, D4, D, PM); gam = fn("Gamma", None, ol.G444, D4, D4, D4)
        dot = fn("dotprod", D, D4, D4, PM); cov = fn("vector_covariant", None, D4, D4, PM)
        null(k, 1.0, C.byref(gf)); gam(G, k, k, dk)
        want["flat_null_k"], want["flat_Gamma"], want["flat_kk"] = np.array(list(k)), np.array(list(dk)), np.array([dot(k, k, C.byref(gf))])
        fn("kerr_connection", None, D, D, D, ol.G444)(a, r, m, G)
        kc = D4(*list(k)); null(kc, 2.0, C.byref(g)); gam(G, kc, kc, dk)
        want["kerr_null_k"], want["kerr_Gamma"] = np.array(list(kc)), np.array(list(dk))
        cov(kc, dk, C.byref(g)); want["kerr_k_cov"] = np.array(list(dk))
        cov(kc, dk, None); want["flat_k_cov"] = np.array(list(dk))
        sp = D4(0.0, 0.3, -0.2, 0.05)
        want["norms"] = np.array([fn("vector_norm", D, D4, PM)(sp, C.byref(g)), L.vector_norm(sp, None), fn("vector_3norm", D, D4)(sp)])
        fn("vector_multiply", None, D4, D)(sp, 2.5); want["multiplied"] = np.array(list(sp))
        Om = 0.7 * fn("OmegaK", D, D, D)(r, a)
        U = D4()
        fn("fourvelocity_zamo", None, PM, D4)(C.byref(g), U); want["fourvelocity_zamo"] = np.array(list(U))
        fn("fourvelocity_azimuthal", None, D, PM, D4)(Om, C.byref(g), U); want["fourvelocity_azimuthal"] = np.array(list(U))
        fn("fourvelocity_radial", None, D, PM, D4)(-0.2, C.byref(g), U); want["fourvelocity_radial"] = np.array(list(U))
        want["fourvelocity_norm"] = np.array([fn("fourvelocity_norm", D, D, D, D, PM)(0.05, 0.01, 0.5 * Om, C.byref(g))])
        fn("fourvelocity", None, D, D, D, PM, D4)(0.05, 0.01, 0.5 * Om, C.byref(g), U); want["fourvelocity"] = np.array(list(U))
        te = lambda x: np.frombuffer(ol.struct_bytes(x), np.float64)[:16].copy()
        fn("tetrad_general", None, PM, D4, PT)(C.byref(g), U, C.byref(t)); want["tetrad_general"] = te(t)
        trad = fn("tetrad_radial", None, PM, D, PT)
        trad(C.byref(g), -0.2, C.byref(t)); want["tetrad_radial"] = te(t)
        trad(C.byref(g), 0.0, C.byref(t)); want["tetrad_radial0"] = te(t)
        want["frequencies"] = np.array([fn("omega_r", D, D, D)(r + 6.0, a), fn("omega_z", D, D, D)(r + 6.0, a),
                                        fn("ell_from_Omega", D, D, PM)(Om, C.byref(g))])
        gd = ol.Geodesic(); e = C.c_int(0)
        v = [float("nan")] * 3
        if fn("geodesic_init_inf", ol.I, D, D, D, D, ol.PG, ol.PI)(deg2rad(65.0), a, 4.0, -3.0, C.byref(gd), C.byref(e)):
            sk = fn("geodesic_position_pol_sign_k_theta", D, ol.PG, D); dms = fn("geodesic_dm_sign", D, ol.PG, D)
            v = [sk(C.byref(gd), 0.4 * gd.Rpc), dms(C.byref(gd), 0.4 * gd.Rpc), sk(C.byref(gd), 1.7 * gd.Rpc)]
        want["sign_k_theta"] = np.array(v)
        epi = fn("elliptic_pi", ol.Cplx, D, D, D)
        z, w = epi(-2.2, 1.8, 0.45), epi(4.0, -0.6, 0.45)
        want["legendre"] = np.array([fn("elliptic_f", D, D, D)(-2.2, 0.45), fn("elliptic_e_sin", D, D, D)(0.8, 0.45),
                                     fn("elliptic_pi_sin", D, D, D, D)(0.8, -0.6, 0.45), z.re, z.im, w.re, w.im])
        E = np.array([0.1, 0.5, 1.0, 3.0, 9.0]); Iv = np.full(5, -1.0)
        bb = fn("blackbody", None, D, D, D, ol.PD, ol.PD, ol.I)
        bb(2.5e6, 1.7, 0.4, E.ctypes.data_as(ol.PD), Iv.ctypes.data_as(ol.PD), 5); want["blackbody"] = Iv.copy()
        bb(0.0, 1.7, 0.4, E.ctypes.data_as(ol.PD), Iv.ctypes.data_as(ol.PD), 5); want["blackbody_T0"] = Iv.copy()
        want["photons"] = np.array([fn("blackbody_photons", D, D, D, D, D)(2.5e6, 1.7, 0.4, 1.0), fn("blackbody_photons_total", D, D, D)(2.5e6, 1.7)])
        want["helpers"] = np.array([1.0, 1.0, 2.0, 0.5])
        assert set(got) == set(want), set(got) ^ set(want)
        for name in want:
            assert_close(got[name], want[name], rtol=1e-9, floor=1e-9, what="%s (a=%g r=%g m=%g)" % (name, a, r, m))


def test_radii_and_disk_report_program(tmp_path, capi, golden):
    """tests/c/disk_dump.c through the scalar API: the radii of example 01, the disk report incl. a set-up by
    luminosity, and disk_nt_dump's table (ref src/sim5disk-nt.c:310-360) against the reference's numbers."""
    exe = str(tmp_path / "dump")
    subprocess.run(["gcc", os.path.join(ROOT, "tests", "c", "disk_dump.c"), os.path.join(ROOT, "src", "sim5lib.c"),
                    "-I", os.path.join(ROOT, "src"), "-o", exe, "-lm", "-O3", "-w", "-fgnu89-inline"], check=True)
    env = dict(os.environ, SIM5GPU_LIB=capi.LIB_PATH)
    p = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.splitlines()
    g = golden("kat_disk_model.npz")
    spin = g["spin"]
    for ln in [l for l in lines if l.startswith("radii")]:
        v = [float(t) for t in ln.split()[1:]]
        i = int(np.argmin(np.abs(spin - v[0])))
        for got, ref in zip(v[1:], (g["r_bh"][i], g["r_ph"][i], g["r_mb"][i], g["r_ms"][i])):
            assert abs(got - ref) <= 1e-12 * max(1.0, abs(ref)), ln
    m0 = [float(t) for t in [l for l in lines if l.startswith("model0")][0].split()[1:]]
    assert m0[0] == g["mdot_3"][0] and abs(m0[1] / g["lumi_3"][0] - 1) < 1e-6 and m0[2] == g["rmin_3"][0]
    m1 = [float(t) for t in [l for l in lines if l.startswith("model1")][0].split()[1:]]
    assert abs(m1[0] / g["mdot_4"][0] - 1) < 1e-6 and abs(m1[1] / g["lumi_4"][0] - 1) < 1e-6
    assert [l for l in lines if l.startswith("zeros")][0].split()[1:] == ["0", "0", "0"]
    # the dump: header of the LAST set-up (luminosity option), then r flux sigma ell 0 0 0 per radius
    assert "# options  = 1" in lines and any(l.startswith("# L        = 3.0000") for l in lines)
    rows = np.array([[float(t) for t in l.split()] for l in lines if l and l[0].isdigit()])
    assert rows.shape[1] == 7 and rows.shape[0] > 100 and (rows[:, 4:] == 0).all()
    assert abs(rows[0, 0] - np.float32(g["rmin_4"][0])) < 1e-6 and np.allclose(rows[1:, 0] / rows[:-1, 0], 1.05, rtol=1e-6)
    orc = ol.Oracle(); orc.disk_nt_setup(10.0, 0.5, 0.3, 0.1, 1)
    want = np.array([[orc.disk_nt_flux(r), orc.disk_nt_sigma(r), orc.disk_nt_ell(r)] for r in rows[:, 0]])
    # "%e" keeps 7 digits of r as well: compare away from the inner edge, where F and Sigma are steep functions of r
    ok = rows[:, 0] > 1.5 * rows[0, 0]
    assert ok.sum() > 100 and np.allclose(rows[ok, 1:4], want[ok], rtol=2e-5, atol=0)
    assert rows[0, 1] == 0 and rows[1, 1] > 0                                 # F = 0 on the inner edge itself


def test_batch_example_program(tmp_path, capi):
    """examples/disk_image_batch.c: the reference example's output format from one library call."""
    exe = str(tmp_path / "batch")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.run(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "disk_image_batch.c"),
                    "-o", exe, "-L", libdir, "-lsim5gpu", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-lm"],
                   check=True)
    p = subprocess.run([exe, "0.5", "60"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = [ln.split() for ln in p.stdout.splitlines() if ln.strip()]
    assert len(rows) == 1280 * 720 and rows[0][:2] == ["0", "0"] and rows[-1][:2] == ["719", "1279"]
    img_g = np.array([float(r[3]) for r in rows]).reshape(720, 1280)
    c = ol.cpu_disk_image("port", 1280, 720, 0.5, 60.0, nthreads=8, full=False)
    # "%e" keeps 7 significant digits
    assert np.allclose(img_g, c["image_g"], rtol=2e-6, atol=0)
    assert "photons: 921600" in p.stderr


def test_sharded_example_program_with_one_rank(tmp_path, capi):
    """examples/disk_image_sharded.c (include/sim5gpu_rccl.h) with a world of ONE rank on the GPU: the C-level multi-GPU
    entry points -- plan, in-place tracing of the rank's mirrored stripes, the begin/end pipeline -- produce the
    reference's image (hit count of BASELINE.md for 1024^2).  More than one rank needs more than one GPU: RCCL refuses
    two ranks on one device, so the gather itself first runs on the driver's multi-GPU node."""
    exe = str(tmp_path / "sharded")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.run(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "disk_image_sharded.c"),
                    "-o", exe, "-L", libdir, "-lsim5gpu_rccl", "-lsim5gpu", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-lm"],
                   check=True)
    p = subprocess.run([exe, "0", "1", str(tmp_path / "id"), "0.998", "70", "1024", "5"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-2000:])
    assert "disk hits 991579" in p.stdout and "rows on rank 0: 1024" in p.stdout, p.stdout


def test_rccl_shard_objects_with_a_world_of_one(capi):
    """The RCCL side with one rank in this process: unique id, ncclCommInitRank, a Shard with a communicator, images
    through begin/end with two in flight, bit-identical to sim5gpu_disk_image; argument errors are reported, not crashed on."""
    from sim5_amd import rccl
    comm = rccl.comm_create(rccl.unique_id(), 0, 1)
    n = 512
    sh = rccl.Shard(comm, 0, 1, n, n)
    want = capi.disk_image(capi.image_desc(n, n, 0.9, deg2rad(60.0)))
    f = [capi.DeviceBuffer(n * n * 4) for _ in range(2)]; g = [capi.DeviceBuffer(n * n * 4) for _ in range(2)]
    d = capi.image_desc(n, n, 0.9, deg2rad(60.0))
    sh.begin(d, f[0].ptr, g[0].ptr)
    sh.begin(d, f[1].ptr, g[1].ptr)
    with pytest.raises(rccl.Sim5GpuRcclError, match="in flight"):
        sh.begin(d, f[0].ptr, g[0].ptr)
    sh.end(); sh.end()
    with pytest.raises(rccl.Sim5GpuRcclError, match="no image in flight"):
        sh.end()
    with pytest.raises(rccl.Sim5GpuRcclError, match="WHOLE-image"):
        sh.image(capi.image_desc(n, n, 0.9, 1.0, y0=0, y1=64), f[0].ptr, g[0].ptr)
    capi.synchronize()
    for b in range(2):
        assert np.array_equal(f[b].to_numpy(np.float32, (n, n)), want["image_f"]) and np.array_equal(g[b].to_numpy(np.float32, (n, n)), want["image_g"])
    sh.destroy()
    rccl.comm_destroy(comm)


def test_a_disk_setup_beside_the_shim_invalidates_its_records(tmp_path, capi):
    """sim5gpu_disk_nt_setup called BESIDE the shim (ctypes, DiskModel_ThinDisk, a C program) changes the
    process-global disk model; the shim's per-ray and look-ahead records carry the library's generation counter
    (sim5gpu_disk_nt_generation) and stop answering disk_nt_flux.  From C (tests/c/shim_generation.c, 34 model changes in the
    middle of a raster walk) and from Python (sim5_amd.sim5lib after capi.disk_nt_setup)."""
    libdir = os.path.dirname(capi.LIB_PATH)
    exe = _cc(tmp_path, "shim_generation.c", "gen", ["-I", os.path.join(ROOT, "include"), "-L", libdir, "-lsim5gpu",
                                                      "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"])
    p = subprocess.run([exe], env=dict(os.environ, SIM5GPU_LIB=capi.LIB_PATH), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.startswith("ok:"), (p.stdout[-1500:], p.stderr[-1500:])
    import sim5_amd.sim5lib as s5
    a, inc = 0.9, deg2rad(60.0)
    s5.disk_nt_setup(10.0, a, 0.1, 0.1, 0)
    gd = s5.geodesic(); err = s5.intp()
    assert s5.geodesic_init_inf(inc, a, 4.0, 3.0, gd, err)
    P = s5.geodesic_find_midplane_crossing(gd, 0)
    r = s5.geodesic_position_rad(gd, P)
    f1 = s5.disk_nt_flux(r)                                     # (answered from the record)
    capi.disk_nt_setup(10.0, a, 1.0, 0.1)                       # beside the shim: ten times the accretion rate
    f2 = s5.disk_nt_flux(r)
    assert f1 > 0 and abs(f2 / f1 - 10.0) < 1e-6, (f1, f2)
    assert f2 == float(capi.disk_nt_flux(np.array([r]))[0])


def test_raytrace_loop_call_by_call_and_with_the_look_ahead(tmp_path, capi):
    """VERDICT r5 missing 3 / item 8: raytrace() ONE CALL AT A TIME through the scalar API (tests/c/raytrace_loop.c = the loop of
    ref src/sim5unittests.c:116-127).  With the shim's look-ahead (sim5gpu_raytrace_record: up to 64 consecutive calls per
    launch, each served after a bit-for-bit check of x, k, *step and *rtd) and with one launch per call the program prints the
    same text; against the SAME program linked with the unmodified reference library: the same number of calls on every ray,
    end states within 1e-6 (r, cos theta, k^r, k^theta; t and phi with a floor of 1)."""
    exe = _cc(tmp_path, "raytrace_loop.c", "rtloop")
    env = dict(os.environ, SIM5GPU_LIB=capi.LIB_PATH)
    outs = []
    for extra in ({}, {"SIM5_SHIM_NO_LOOKAHEAD": "1"}):
        p = subprocess.run([exe, "0.9", "60", "4"], env=dict(env, **extra), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(p.stdout)
    body = [[ln for ln in o.splitlines() if not ln.startswith("#")] for o in outs]
    assert body[0] == body[1] and len(body[0]) >= 2, "the look-ahead run differs from the call-by-call run"
    rate = [float([ln for ln in o.splitlines() if ln.startswith("# raytrace loop")][0].split()[10]) for o in outs]
    print("raytrace() call by call through the scalar API: %.3e calls/s with the look-ahead, %.3e with one launch per call" % tuple(rate))
    assert rate[0] > 1.5 * rate[1]
    if ol.have_reference():
        rdir = os.path.dirname(ol.REF_SO)
        ref_exe = str(tmp_path / "rtloop_ref")
        subprocess.run(["gcc", os.path.join(ROOT, "tests", "c", "raytrace_loop.c"), "-I", HOST, "-o", ref_exe, "-L", rdir, "-lsim5ref",
                        "-Wl,-rpath," + rdir, "-lm", "-O3", "-w", "-fgnu89-inline"], check=True)
        r = subprocess.run([ref_exe, "0.9", "60", "4"], capture_output=True, text=True, timeout=600)
        ref = [[float(v) for v in ln.split()] for ln in r.stdout.splitlines() if not ln.startswith("#")]
        got = [[float(v) for v in ln.split()] for ln in body[0]]
        assert len(got) == len(ref) and [g[0] for g in got] == [w[0] for w in ref]
        for g, w in zip(got, ref):
            assert g[1] == w[1], ("raytrace() calls", g[1], w[1])
            for c, floor in ((2, 1.0), (3, 0.0), (4, 1e-2), (5, 1.0), (6, 1e-2), (7, 1e-4)):
                assert abs(g[c] - w[c]) <= 1e-6 * max(abs(w[c]), floor), (c, g[c], w[c])
