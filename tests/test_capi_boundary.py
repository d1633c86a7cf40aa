"""The C-ABI boundary without a GPU: the library loads, exports every symbol include/sim5gpu.h
declares, keeps the SIM5 struct layouts, and fails loudly (no CPU fallback) when no device exists."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "sim5gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sim5gpu_[a-zA-Z0-9_]+)\s*\(", src)))


def test_header_compiles_as_c():
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write('#include "sim5gpu.h"\n'
                           '_Static_assert(sizeof(sim5gpu_geodesic)==240, "geodesic");\n'
                           '_Static_assert(sizeof(sim5gpu_metric)==64, "metric");\n'
                           '_Static_assert(sizeof(sim5gpu_tetrad)==192, "tetrad");\n'
                           '_Static_assert(sizeof(sim5gpu_raytrace_data)==144, "rtd");\n'
                           '_Static_assert(sizeof(sim5gpu_stokes)==40, "stokes");\n'
                           '_Static_assert(__builtin_offsetof(sim5gpu_geodesic, l)==40, "l");\n'
                           '_Static_assert(__builtin_offsetof(sim5gpu_geodesic, nrr)==120, "nrr");\n'
                           '_Static_assert(__builtin_offsetof(sim5gpu_geodesic, Rpc)==176, "Rpc");\n'
                           '_Static_assert(__builtin_offsetof(sim5gpu_raytrace_data, dk)==64, "dk");\n'
                           '_Static_assert(__builtin_offsetof(sim5gpu_raytrace_data, error)==136, "error");\n'
                           'int main(void){return 0;}\n')
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c,
                        "-o", os.path.join(td, "t")], check=True)


def test_library_exports_every_declared_symbol(capi):
    names = declared_functions()
    assert len(names) >= 50
    lib = C.CDLL(capi.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_layouts_match_header(capi):
    assert capi.GEODESIC_DTYPE.itemsize == 240 and capi.GEODESIC_DTYPE.fields["Rpc"][1] == 176
    assert capi.RAYTRACE_DTYPE.itemsize == 144 and capi.RAYTRACE_DTYPE.fields["error"][1] == 136
    assert capi.METRIC_DTYPE.itemsize == 64 and capi.TETRAD_DTYPE.itemsize == 192 and capi.STOKES_DTYPE.itemsize == 40


def test_fails_loudly_without_gpu(capi):
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(capi.Sim5GpuError, match="no CPU fallback"):
        capi.gfactorK([6.0], 0.5, 1.0)
    with pytest.raises(capi.Sim5GpuError):
        capi.disk_image(capi.image_desc(16, 16, 0.5, 1.0))
    with pytest.raises(capi.Sim5GpuError):
        capi.DeviceBuffer(16)


def test_argument_validation_needs_no_gpu(capi):
    rc = capi._lib.sim5gpu_gfactorK(C.c_size_t(4), None, None, None, None)
    assert rc == -3 and b"NULL" in capi._lib.sim5gpu_last_error()
    bad = capi.image_desc(0, 0, 0.5, 1.0)
    f = (C.c_float * 4)()
    assert capi._lib.sim5gpu_disk_image_host(C.byref(bad), f, f, None) == -3
    # unknown option bits are an argument error; the luminosity option (1) needs the device for its Simpson integrals
    assert capi._lib.sim5gpu_disk_nt_setup(C.c_double(10), C.c_double(.5), C.c_double(.1), C.c_double(.1), C.c_int(2)) == -3
    if capi.device_count() == 0:
        assert capi._lib.sim5gpu_disk_nt_setup(C.c_double(10), C.c_double(.5), C.c_double(.1), C.c_double(.1), C.c_int(1)) == -1


def test_product_does_not_reach_into_oracle():
    """No file of the product tree may include, link or import anything under oracle/."""
    bad = []
    for base in ("sim5_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(ROOT, base)):
            if "_build" in dp or dp.endswith("/lib"):
                continue
            for f in fn:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".c", ".cpp")) or f == "Makefile":
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"oracle[/.]|oraclelib|liboracle|sim5ref", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


# ---- the SIM5 scalar boundary against the inventory of the reference's headers ------------------------------------------
# Prototypes of the cited headers that are NOT served, each with the reason (explicit list).
OUT_OF_SCOPE = {
    "blackbody_photon_energy_random": "Monte-Carlo sampler on the RNG (SURVEY.md 2 row 15: RNG out of scope)",
    "sim5seed": "RNG", "sim5rand": "RNG", "sim5urand": "RNG",
    "cartesian2spherical1": "cartesian helper, no caller on the path", "cartesian2spherical2": "cartesian helper, no caller on the path",
    "quadratic_eq": "dead code in the reference: the quartic is solved in closed form inside geodesic_priv_R_roots (SURVEY.md 2 row 4)",
    "cubic_eq": "dead code (as quadratic_eq)", "quartic_eq": "dead code: zero callers", "quartic_eq_c": "dead code",
    "sort_roots_re": "helper of the dead quartic_eq", "sort_mix": "helper of the dead quartic_eq", "sort_mix2": "helper of the dead quartic_eq",
}


def _inventory():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_prototypes.json")))["prototypes"]


def _list_prototypes_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("list_prototypes", os.path.join(ROOT, "oracle", "list_prototypes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def host_shim_so(tmp_path_factory):
    """sim5_amd/host/sim5lib.c as a shared object (plain C, the reference's example flags): resolvable symbols"""
    import subprocess
    so = str(tmp_path_factory.mktemp("shim") / "libsim5shim.so")
    host = os.path.join(ROOT, "sim5_amd", "host")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O2", "-w", "-fgnu89-inline", "-I", host, os.path.join(host, "sim5lib.c"),
                    "-o", so, "-lm"], check=True)
    return so


def test_host_shim_serves_every_prototype_of_the_cited_headers(host_shim_so):
    """Every public prototype of the nine reference headers the boundary cites (inventory: oracle/list_prototypes.py ->
    tests/golden/reference_prototypes.json, 140 names with their type signatures) is declared by sim5_amd/host/sim5lib.h
    with the same types and defined by sim5lib.c, except the names of OUT_OF_SCOPE."""
    lp = _list_prototypes_module()
    inv = _inventory()
    assert len(inv) >= 138
    ours = {p["name"]: p for p in lp.prototypes(os.path.join(ROOT, "sim5_amd", "host", "sim5lib.h"), marked=False)}
    lib = C.CDLL(host_shim_so)
    missing, differ, unresolved = [], [], []
    for p in inv:
        if p["name"] in OUT_OF_SCOPE:
            assert p["name"] not in ours, "%s is served: take it off the out-of-scope list" % p["name"]
            continue
        if p["name"] not in ours:
            missing.append("%s (%s:%d)" % (p["name"], p["header"], p["line"]))
        elif ours[p["name"]]["signature"] != p["signature"]:
            differ.append((p["name"], p["signature"], ours[p["name"]]["signature"]))
        if not hasattr(lib, p["name"]):
            unresolved.append(p["name"])
    assert not missing, missing
    assert not differ, differ
    assert not unresolved, unresolved
    assert set(OUT_OF_SCOPE) <= {p["name"] for p in inv}
    # in the build container the committed inventory is re-derived from the reference's headers
    if os.path.exists(os.path.join(lp.REF, "src", "sim5kerr.h")):
        fresh = []
        for h in lp.HEADERS:
            for p in lp.prototypes(os.path.join(lp.REF, "src", h)):
                if not any(q["name"] == p["name"] for q in fresh):
                    fresh.append(dict(p, header="src/" + h))
        key = lambda ps: [(p["name"], p["signature"], p["header"], p["line"]) for p in ps]
        assert key(fresh) == key(inv)


def test_host_side_helpers_match_the_reference(host_shim_so, golden):
    """The helpers the shim answers itself -- they move, compare or reorder values, no ray arithmetic: ensure_range,
    sort_roots, the angle reductions, sim5round, factorial, the complex accessors, vector_set/copy/multiply -- against
    the reference's outputs (kat_boundary.npz); none of them needs the GPU library."""
    g = golden("kat_boundary.npz")
    L = C.CDLL(host_shim_so)
    D, PD = C.c_double, C.POINTER(C.c_double)

    class Cx(C.Structure):
        _fields_ = [("re", D), ("im", D)]
    L.ensure_range.argtypes = [PD, D, D, D]; L.ensure_range.restype = C.c_int
    for v, acc, ok, out in zip(g["er_val"], g["er_acc"], g["er_ok"], g["er_out"]):
        x = D(v)
        assert L.ensure_range(C.byref(x), -1.0, 1.0, acc) == ok and x.value == out
    PC = C.POINTER(Cx)
    L.sort_roots.argtypes = [C.POINTER(C.c_int), PC, PC, PC, PC]; L.sort_roots.restype = None
    for zin, zout, nre in zip(g["roots_in"], g["roots_sorted"], g["roots_nreal"]):
        z = [Cx(*zin[j]) for j in range(4)]; s = C.c_int(-1)
        L.sort_roots(C.byref(s), C.byref(z[0]), C.byref(z[1]), C.byref(z[2]), C.byref(z[3]))
        assert s.value == nre and np.array_equal(np.array([(q.re, q.im) for q in z]), zout)
    for name, key in (("reduce_angle_pi", "reduce_pi"), ("reduce_angle_2pi", "reduce_2pi")):
        fn = getattr(L, name); fn.argtypes = [D]; fn.restype = D
        assert np.array_equal(np.array([fn(x) for x in g["angles"]]), g[key])
    L.sim5round.argtypes = [D]; L.sim5round.restype = C.c_long
    assert np.array_equal(np.array([L.sim5round(x) for x in g["round_in"]]), g["round_out"])
    L.factorial.argtypes = [C.c_long]; L.factorial.restype = C.c_long
    assert np.array_equal(np.array([L.factorial(k) for k in range(15)]), g["factorial"])
    L.makeComplex.argtypes = [D, D]; L.makeComplex.restype = Cx
    L.nullComplex.restype = Cx
    L.sim5creal.argtypes = [Cx]; L.sim5creal.restype = D
    L.sim5cimag.argtypes = [Cx]; L.sim5cimag.restype = D
    z = L.makeComplex(1.5, -2.25)
    assert (z.re, z.im) == (1.5, -2.25) and L.sim5creal(z) == 1.5 and L.sim5cimag(z) == -2.25
    z0 = L.nullComplex(); assert (z0.re, z0.im) == (0.0, 0.0)
    D4 = D * 4
    L.vector_set.argtypes = [D4, D, D, D, D]; L.vector_copy.argtypes = [D4, D4]; L.vector_multiply.argtypes = [D4, D]
    v = D4(); w = D4()
    L.vector_set(v, 1.0, 2.0, 3.0, 4.5); L.vector_copy(v, w); L.vector_multiply(w, 0.1)
    assert list(v) == [1.0, 2.0, 3.0, 4.5] and list(w) == [1.0 * 0.1, 2.0 * 0.1, 3.0 * 0.1, 4.5 * 0.1]
    L.geodesic_position.argtypes = [C.c_void_p, D, D4]; L.geodesic_position.restype = None
    L.geodesic_position(None, 1.0, v)                       # the reference's empty stub: nothing is touched
    assert list(v) == [1.0, 2.0, 3.0, 4.5]


# ---- the multi-GPU library (include/sim5gpu_rccl.h) ----------------------------------------------------------------------
def test_rccl_library_exports_every_declared_symbol_and_fails_loudly(capi):
    src = open(os.path.join(ROOT, "include", "sim5gpu_rccl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(sim5gpu_[a-zA-Z0-9_]+)\s*\(", src)))
    assert len(names) >= 10
    from sim5_amd import rccl
    lib = C.CDLL(rccl.LIB_PATH)
    assert not [n for n in names if not hasattr(lib, n)]
    # the base library keeps no RCCL dependency
    import subprocess
    needed = subprocess.run(["readelf", "-d", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in needed.lower()
    if capi.device_count() == 0:
        with pytest.raises(rccl.Sim5GpuRcclError, match="no CPU fallback"):
            rccl.Shard(None, 0, 1, 64, 64)


def test_shard_plan_is_the_python_dealing_rule(capi):
    """sim5gpu_shard_plan (C, host arithmetic) against sim5_amd/sharding.py, the rule the gloo tests and the randomised
    campaign (tests/tools/fuzz_stripes.py) check: rows per rank, the band, and the share's job description row by row"""
    from sim5_amd import rccl, sharding
    for (ny, world, dealt) in [(4096, 8, 0), (4096, 2, 640), (4096, 4, 1024), (1001, 3, 192), (8192, 8, 2560), (129, 2, 0), (65, 1, 0)]:
        d = capi.image_desc(ny, ny, 0.9, 1.0)
        dd = None if dealt == 0 else dealt
        total = 0
        for r in range(world):
            rows, band, share = rccl.shard_plan(d, r, world, dealt)
            assert rows == sharding.rank_rows(ny, r, world, dealt=dd if world > 1 else None), (ny, world, dealt, r)
            want_band = (sharding.root_band(ny, dd) if world > 1 else None) or (0, 0)
            assert band == tuple(want_band)
            own = [y for (y0, y1) in sharding.stripes_for_rank(ny, r, world, dealt=dd if world > 1 else None) for y in range(y0, y1)]
            if own:
                assert capi.image_row_map(share).tolist() == own
                assert bool(share.flags & capi.IMG_INPLACE) == (r == 0) and (share.flags & capi.IMG_MIRROR)
            total += rows
        assert total == ny


def test_default_field_of_view_is_the_reference_r_ms_bit_for_bit(capi):
    """rmax = r_ms(a) + 8 (ref disk-image.c:41-42) scales every alpha and beta of an image: the library's host-side r_ms must be
    the reference's to the last bit for EVERY spin (round 5: 3.*sqr(a) had been written (3 a) a -- one ulp off for random spins,
    found by the randomised campaign on the central column of one image in 1 500).  Against the unmodified reference, 20 000
    random spins and the usual ones."""
    import ctypes as C
    import oraclelib as ol
    if not ol.have_reference():
        pytest.skip("oracle/_ref/libsim5ref.so not present")
    L = C.CDLL(ol.REF_SO)
    L.r_ms.restype = C.c_double; L.r_ms.argtypes = [C.c_double]
    rng = np.random.default_rng(7)
    spins = np.concatenate([[0.0, 1e-5, 0.3, 0.5, 0.7, 0.9, 0.998, 0.9999, 0.999999], rng.uniform(0.0, 0.999999, 20000)])
    for a in spins:
        rmax, rms, _, _ = capi.image_view(capi.image_desc(8, 8, float(a), 1.0))
        ref = L.r_ms(float(a))
        assert rms == ref and rmax == ref + 8.0, (float(a), rms, ref)
    # ... and sin i / cos i are the reference binary's: cos_i and l = -alpha sin i of the record its geodesic_init_inf makes (gcc
    # merges its sin(i), cos(i) into one sincos() call, whose cosine is not cos()'s in the last bit for every argument)
    init = L.geodesic_init_inf
    init.restype = C.c_int; init.argtypes = [C.c_double] * 4 + [C.c_void_p, C.c_void_p]
    g = (C.c_double * 30)(); err = C.c_int(0)
    differs_from_plain_cos = 0
    for inc in np.concatenate([np.radians([10, 20, 30, 40, 50, 60, 70, 80]), rng.uniform(0.02, 1.55, 20000)]):
        _, _, si, ci = capi.image_view(capi.image_desc(8, 8, 0.5, float(inc)))
        init(float(inc), 0.5, -1.0, 3.0, g, C.byref(err))
        assert ci == g[4] and si == g[5], (float(inc), ci, g[4], si, g[5])          # cos_i; l = -(-1) sin i
        differs_from_plain_cos += int(ci != math.cos(float(inc)))
    print("inclinations where sincos()'s cosine is not cos()'s: %d of 20008" % differs_from_plain_cos)


def test_shipped_kernels_are_one_source_one_build():
    """two readers must be able to tell which kernel is the tested one.  (i) The library the product loads is
    compiled with the strict / fast pair and nothing else: no -DS5_* besides -DS5_FAST=0|1 in the command line recorded next
    to every object (experiment flags reach variant builds only, sim5_amd/build.py).  (ii) The image kernels' sources carry
    no knock-out / ablation / A/B preprocessor branch any more: the only macros their conditionals test are the variant
    (S5_FAST and what s5_config.hpp derives from it), the assembly-comment marker of tests/tools/isa_spill_sites.py and the
    march kernel's one instrumentation switch -- in EVERY file of sim5_amd/csrc."""
    import glob
    import re
    from sim5_amd import capi      # (builds the library if it is not there)
    csrc = os.path.join(ROOT, "sim5_amd", "csrc")
    cmds = glob.glob(os.path.join(csrc, "_build", "*.o.cmd"))
    assert len(cmds) >= 16, cmds
    for f in cmds:
        flags = re.findall(r"-D(S5_\w+)(?:=(\S+))?", open(f).read())
        assert [n for n, _ in flags] == ["S5_FAST"] and flags[0][1] in ("0", "1"), (f, flags)
    allowed = {"S5_FAST", "S5_F_SQRTDIV", "S5_F_RF7", "S5_F_AGMK", "S5_F_LIBM", "S5_FPC_MASK", "S5_RPC_ADD", "S5_ISA_MARKS",
               "S5_WAVE_VOTES"}         # (set by k_surface.hip in its own source: s5_math.hpp)
    # (the march kernel keeps ONE instrumentation switch, S5_TORUS_DEBUG: wave timelines and batch-fill counters for
    # tests/tools/torus_occupancy.py, built as a variant library; S5_MARCH_LEAN is S5_FAST by another name)
    allowed |= {"S5_TORUS_DEBUG", "S5_MARCH_LEAN"}
    sources = [f for f in os.listdir(csrc) if f.endswith((".hpp", ".hip"))]
    assert len(sources) >= 28
    for name in sources:
        for ln in open(os.path.join(csrc, name)):
            t = ln.strip()
            if re.match(r"#\s*(if|ifdef|ifndef|elif)\b", t):
                used = set(re.findall(r"\bS5_\w+", t))
                assert used <= allowed, (name, t)


def test_no_register_spills_in_the_image_kernels():
    """The whole-image kernels of the fast variant run without scratch memory: no spilled VGPR, no private segment.  (A build
    of disk_image_mirror_kernel that spilled ONE double under its 128-register cap produced wrong pixels in some launches
    and a memory fault in others -- DESIGN.md 4, round 3 -- so a spill there is a build error, not a tuning matter.)"""
    from sim5_amd import capi
    from sim5_amd.codeobj import kernel_metadata
    meta = kernel_metadata(capi.LIB_PATH)
    image = {k: v for k, v in meta.items() if "s5f" in k and "disk_image" in k}
    assert len(image) >= 11, sorted(meta)
    for k, v in image.items():
        assert v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (k, v)
    # the production kernel (every job without full-precision planes: bench.py, the sharded path): its parameters are read from
    # the argument segment where they are used.  What a spilled scalar register COSTS is the lane moves (v_writelane /
    # v_readlane) on the executed path; the metadata's sgpr_spill_count counts reserved slots.  Of the ~70 lane moves of the
    # kernel all but a handful sit in the cold re-trace a few rays per million take: the hot path -- three class instantiations
    # -- holds at most 12 (tests/tools/isa_spill_sites.py compiles the kernel with region marks and says where they are)
    from sim5_amd.codeobj import lane_moves
    jobs = {k: v for k, v in image.items() if "disk_image_jobs_kernel" in k}
    assert len(jobs) == 2, sorted(image)
    moves = lane_moves(capi.LIB_PATH, "disk_image_jobs_kernel")
    assert len(moves) == 2 and all(w + r <= 90 for (w, r) in moves.values()), moves
    import os, subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "isa_spill_sites.py")
    res = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    march = [v for k, v in meta.items() if "torus_pool_kernel" in k]
    assert len(march) == 2 and all(v["vgpr_spill_count"] == 0 for v in march), march          # both variants
    # nothing else of the library spills a register either, but for the set-up kernel of the surface search in its strict
    # variant (293 registers with its accumulation registers; once per ray, checked against the CPU every round)
    spilling = sorted(k for k, v in meta.items() if v.get("vgpr_spill_count", 0) > 0)
    assert all("surface_setup_kernel" in k and "s5f" not in k for k in spilling), spilling
    # the occupancy the launchers count on: four waves per SIMD for the unpolarized kernels, three for the polarized pairs
    for k, v in image.items():
        cap = 168 if "polarized" in k else 128
        assert v["vgpr_count"] <= cap, (k, v)
