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


def _stub(tmp_path, tsan=False):
    """tests/c/stub_sim5gpu.c as a shared library: a TEST DOUBLE of libsim5gpu.so with made-up arithmetic -- what it is good
    for is running the HOST side of the scalar API (records, look-ahead, lazy symbol resolution) without a GPU"""
    so = str(tmp_path / "libstub.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-g", os.path.join(ROOT, "tests", "c", "stub_sim5gpu.c"), "-o", so, "-lm"]
                   + (["-fsanitize=thread"] if tsan else []), check=True)
    return so


def test_scalar_api_host_side_is_clean_under_thread_sanitizer(tmp_path):
    """tests/c/shim_threads.c -- eight host threads sharing an image through the scalar API, batch calls in
    between -- with the host shim compiled by gcc -fsanitize=thread over the test double: no data race reported (the lazily
    resolved function pointers, the library handle, the record / look-ahead switches and the per-thread records), and the
    text of the eight-thread run is the single-threaded run's."""
    probe = tmp_path / "tsan_probe.c"
    probe.write_text("int main(void) { return 0; }\n")
    if subprocess.run(["gcc", "-fsanitize=thread", str(probe), "-o", str(tmp_path / "tsan_probe")], capture_output=True).returncode != 0:
        pytest.skip("this gcc has no ThreadSanitizer runtime")
    so = _stub(tmp_path, tsan=True)
    exe = str(tmp_path / "threads_tsan")
    subprocess.run(["gcc", "-fsanitize=thread", "-g", "-O1", "-w", "-fgnu89-inline", os.path.join(ROOT, "tests", "c", "shim_threads.c"),
                    os.path.join(HOST, "sim5lib.c"), "-I", HOST, "-I", os.path.join(ROOT, "include"), "-o", exe,
                    "-L", str(tmp_path), "-lstub", "-Wl,-rpath," + str(tmp_path), "-lpthread", "-lm"], check=True)
    env = dict(os.environ, SIM5GPU_LIB=so, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    outs = []
    for threads in ("1", "8"):
        p = subprocess.run([exe, "0.9", "60", "48", threads], env=env, capture_output=True, text=True, timeout=600)
        assert "ThreadSanitizer" not in p.stderr, p.stderr[-3000:]
        assert p.returncode == 0, (p.returncode, p.stderr[-2000:])
        outs.append(p.stdout)
    assert outs[0] == outs[1] and outs[0].count("\n") == 48 * 48 + 1


def test_a_disk_setup_beside_the_shim_invalidates_its_records(tmp_path):
    """the generation counter of the disk model lives in libsim5gpu; a sim5gpu_disk_nt_setup made by
    anybody (here: directly, beside sim5lib.c) stops the shim's records from answering disk_nt_flux (tests/c/shim_generation.c,
    here over the test double; tests/test_gpu_host_shim.py runs it over the real library)."""
    so = _stub(tmp_path)
    exe = str(tmp_path / "gen")
    subprocess.run(["gcc", "-O2", "-w", "-fgnu89-inline", os.path.join(ROOT, "tests", "c", "shim_generation.c"), os.path.join(HOST, "sim5lib.c"),
                    "-I", HOST, "-I", os.path.join(ROOT, "include"), "-o", exe, "-L", str(tmp_path), "-lstub", "-Wl,-rpath," + str(tmp_path), "-lm"],
                   check=True)
    p = subprocess.run([exe], env=dict(os.environ, SIM5GPU_LIB=so), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.startswith("ok:"), (p.stdout[-1500:], p.stderr[-1500:])
