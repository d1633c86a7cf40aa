"""Host-side SIM5 API shim (sim5_amd/host): compiles as plain C with the reference's flags, the
reference's own example links against it unchanged (when the reference tree is around), and without a
GPU the program stops with an error instead of computing on the CPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "sim5_amd", "host")
REF_EXAMPLE = "/root/reference/examples/04-disk-image-eqplane/disk-image.c"
FLAGS = ["-O3", "-w", "-fgnu89-inline"]          # ref examples/04-disk-image-eqplane/Makefile:4


def test_shim_compiles_with_reference_flags(tmp_path):
    subprocess.run(["gcc", "-c", os.path.join(HOST, "sim5lib.c"), "-I", HOST, "-o", str(tmp_path / "s.o")] + FLAGS,
                   check=True)


def test_probe_and_batch_example_build(tmp_path, capi):
    subprocess.run(["gcc", os.path.join(ROOT, "tests", "c", "shim_probe.c"), os.path.join(HOST, "sim5lib.c"),
                    "-I", HOST, "-o", str(tmp_path / "probe"), "-lm"] + FLAGS, check=True)
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.run(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "disk_image_batch.c"),
                    "-o", str(tmp_path / "batch"), "-L", libdir, "-lsim5gpu", "-Wl,-rpath," + libdir,
                    "-Wl,-rpath-link,/opt/rocm/lib", "-lm"], check=True)
    if capi.device_count() == 0:
        env = dict(os.environ, SIM5GPU_LIB=capi.LIB_PATH)
        p = subprocess.run([str(tmp_path / "probe"), "0.5", "60", "4"], env=env, capture_output=True, text=True)
        assert p.returncode != 0 and "no CPU fallback" in p.stderr
        p = subprocess.run([str(tmp_path / "batch"), "0.5", "60"], capture_output=True, text=True)
        assert p.returncode != 0 and "no CPU fallback" in p.stderr


@pytest.mark.skipif(not os.path.exists(REF_EXAMPLE), reason="reference tree not present on this box")
def test_reference_example_links_unchanged(tmp_path, capi):
    exe = str(tmp_path / "disk-image")
    subprocess.run(["gcc", "-I", HOST, REF_EXAMPLE, os.path.join(HOST, "sim5lib.c"), "-o", exe, "-lm"] + FLAGS,
                   check=True)
    if capi.device_count() == 0:
        env = dict(os.environ, SIM5GPU_LIB=capi.LIB_PATH)
        p = subprocess.run([exe, "0.5", "60"], env=env, capture_output=True, text=True)
        assert p.returncode != 0 and "no CPU fallback" in p.stderr
