"""The reference's OWN callers over this project's boundary, unchanged (build container only: they are read from
/root/reference where they lie and never copied into the repo; on a box without the reference tree these tests skip).

* python/sim5diskraytrace.py + python/sim5diskmodel.py imported as they are, with `sim5lib` = sim5_amd.sim5lib
  (the SWIG-name module).  Without a GPU its C-ABI calls are served by the CPU oracle (tests/oracle_capi.py), which
  checks the module's glue (names, argument order, pointer helpers, struct members) against the golden images the
  same classes produced over the reference library (tests/golden/py_diskraytrace.npz).
* examples/04-disk-image-eqplane and examples/01-kerr-spacetime built by their own Makefiles, default relative
  layout (../../src, ../../lib), zero edits, against this repo's src/ and lib/.
"""
import importlib
import logging
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "python")), reason="reference tree not present on this box")


class _OracleShim:
    """The entry points sim5_amd/sim5lib.py takes from libsim5shim.so (ctypes signatures: pointers as integers), answered by
    the CPU checker: test infrastructure for the container without a GPU"""
    def __init__(self, oc):
        self.oc = oc

    @staticmethod
    def _rec(ptr):
        import ctypes as C
        import sim5_amd.capi as capi
        return np.frombuffer((C.c_char * 240).from_address(ptr), dtype=capi.GEODESIC_DTYPE)

    def geodesic_init_inf(self, i, a, alpha, beta, gptr, err_ref):
        rec, err, ok = self.oc.geodesic_init_inf(i, a, [alpha], beta)
        self._rec(gptr)[:] = rec
        err_ref._obj.value = int(err[0])
        return int(ok[0])

    def geodesic_find_midplane_crossing(self, gptr, order):
        return float(self.oc.geodesic_find_midplane_crossing(self._rec(gptr), order)[0])

    def geodesic_position_rad(self, gptr, P):
        return float(self.oc.geodesic_position_rad(self._rec(gptr), P)[0])

    def gfactorK(self, r, a, l):
        return float(self.oc.gfactorK([r], a, l)[0])

    def disk_nt_flux(self, r):
        return float(self.oc.disk_nt_flux([r])[0])

    def disk_nt_setup(self, M, a, mdot, alpha, options):
        self.oc.disk_nt_setup(M, a, mdot, alpha, options)
        return 0


@needs_ref
def test_reference_python_raytracer_runs_unchanged_over_sim5lib_module(golden, monkeypatch):
    import oracle_capi
    import sim5_amd.sim5lib as s5
    monkeypatch.setattr(s5, "_c", oracle_capi)                  # CPU stand-in for the C-ABI (no GPU in this container)
    monkeypatch.setattr(s5, "_shim", lambda: _OracleShim(oracle_capi))     # ... and for the C host shim the image loop's calls go through
    monkeypatch.setitem(sys.modules, "sim5lib", s5)
    monkeypatch.setattr(np, "float", float, raising=False)      # the reference predates numpy 1.24 (python/sim5diskraytrace.py:154)
    monkeypatch.syspath_prepend(os.path.join(REF, "python"))
    for name in ("sim5diskmodel", "sim5diskraytrace"):
        sys.modules.pop(name, None)
    logging.disable(logging.CRITICAL)
    try:
        model = importlib.import_module("sim5diskmodel")
        rtmod = importlib.import_module("sim5diskraytrace")
        assert model.__file__.startswith(REF) and rtmod.__file__.startswith(REF)
        g = golden("py_diskraytrace.npz")
        for ci, (a, inc) in enumerate(g["cases"]):
            disk = model.DiskModel_ThinDisk(10.0, float(a), 0.1, 0.1)         # calls disk_nt_setup / mdot / lumi / r_min
            assert disk.mdot == np.float32(0.1) and disk.lumi > 0 and disk.r_min > 1.0
            assert disk.sigma(10.0) > 0
            rt = rtmod.DiskRaytrace(10.0, float(a), 10.0, disk, None)
            N = g["img%d_flux" % ci].shape[0]
            img = rt.image(float(inc), float(g["rmax%d" % ci][0]), N)
            for k in ("flux", "gfactor", "mue", "T", "R", "H"):
                ref = g["img%d_%s" % (ci, k)]
                got = np.array(img[k], dtype=np.float64)
                assert np.array_equal(np.isnan(ref), np.isnan(got)), (ci, k)
                m = ~np.isnan(ref)
                if m.any():
                    scale = np.maximum(np.abs(ref[m]), 1e-9 * np.abs(ref[m]).max() + 1e-300)
                    assert np.max(np.abs(got[m] - ref[m]) / scale) < 1e-12, (ci, k)
    finally:
        logging.disable(logging.NOTSET)
        for name in ("sim5diskmodel", "sim5diskraytrace"):
            sys.modules.pop(name, None)


def _tree_with_example(tmp_path, example):
    """<tmp>/src, <tmp>/lib = copies of this repo's (three tiny include files each), <tmp>/sim5_amd -> the repo's
    package (what they include), <tmp>/examples/<example> = the reference's directory as it is."""
    for d in ("src", "lib"):
        shutil.copytree(os.path.join(ROOT, d), tmp_path / d)
    os.symlink(os.path.join(ROOT, "sim5_amd"), tmp_path / "sim5_amd")
    dst = tmp_path / "examples" / example
    shutil.copytree(os.path.join(REF, "examples", example), dst)
    return dst


@needs_ref
@pytest.mark.parametrize("example,target,args", [("04-disk-image-eqplane", "disk-image", ["0.5", "60"]),
                                                 ("01-kerr-spacetime", "kerr-orbits", [])])
def test_reference_example_builds_with_its_own_makefile(tmp_path, capi, example, target, args):
    d = _tree_with_example(tmp_path, example)
    r = subprocess.run(["make", "-C", str(d)], capture_output=True, text=True)        # no SIM5LIB override, no edits
    assert r.returncode == 0, r.stderr[-3000:]
    exe = d / target
    assert exe.exists()
    if capi.device_count() == 0:
        env = dict(os.environ, SIM5GPU_LIB=capi.LIB_PATH)
        p = subprocess.run([str(exe)] + args, env=env, capture_output=True, text=True, cwd=str(d))
        assert p.returncode != 0 and "no CPU fallback" in p.stderr
