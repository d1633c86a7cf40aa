"""The C-ABI boundary without a GPU: the library loads, exports every symbol include/sim5gpu.h
declares, keeps the SIM5 struct layouts, and fails loudly (no CPU fallback) when no device exists."""
import ctypes as C
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
