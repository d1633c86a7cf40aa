"""Randomised campaign over the KNOWN-ANSWER functions (run on the GPU box; needs oracle/_ref/libsim5ref.so, which travels):
the generators of oracle/gen_golden.py are run again with another seed and `scale` times their record counts (elliptic:
1 500 -> 150 000 argument sets, Kerr / tetrads / photon momentum: 600 -> 60 000 points, polarization: 400 -> 40 000, vectors,
geodesic records: 110 -> 110 x scale rays per spin and inclination), into a directory of their own, and the suite's own
known-answer tests (tests/test_gpu_kat.py and the raytrace API test) are pointed at it (SIM5_GOLDEN_DIR): the same
comparisons and tolerances, the live reference's answers for fresh inputs.
usage: python tests/tools/fuzz_kat.py [scale] [seed]"""
import os, re, shutil, subprocess, sys, types
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tempfile
out = os.path.join(tempfile.gettempdir(), "sim5_fuzz_kat_fixtures")          # (tens of MB at scale 100: not under gpurun_out/)
shutil.rmtree(out, ignore_errors=True)
shutil.copytree(os.path.join(root, "tests", "golden"), out)
src = open(os.path.join(root, "oracle", "gen_golden.py")).read()
# the record counts of the generators that are plain `n = <literal>` lines, and the rays per (spin, inclination) of the geodesic set
src, k = re.subn(r"^(    n = )(\d+)$", lambda m: m.group(1) + str(int(m.group(2)) * scale), src, flags=re.M)
src, k2 = re.subn(r"for _ in range\(110\):", "for _ in range(%d):" % (110 * min(scale, 20)), src)
assert k >= 5 and k2 == 1, (k, k2)
src, k3 = re.subn(r"for _ in range\(300\)\]", "for _ in range(%d)]" % (300 * min(scale, 50)), src)        # arguments per azimuth integral
src, k4 = re.subn(r"for _ in range\(160\):", "for _ in range(%d):" % (160 * min(scale, 20)), src)       # rays per (spin, inclination), azimuth / time delay
src, k5 = re.subn(r"for j in range\(8\):", "for j in range(%d):" % (8 * min(scale, 4)), src)             # rays of the raytrace() step sequences
assert k3 == 1 and k4 == 1 and k5 == 1, (k3, k4, k5)
src = re.sub(r"np\.random\.default_rng\((\d+)\)", lambda m: "np.random.default_rng(%d)" % (int(m.group(1)) + seed), src)
mod = types.ModuleType("gen_golden_campaign")
mod.__file__ = os.path.join(root, "oracle", "gen_golden.py")
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "oracle"))
exec(compile(src, mod.__file__, "exec"), mod.__dict__)
mod.OUT = out
import numpy as np
import oraclelib as ol
assert ol.have_reference(), "oracle/_ref/libsim5ref.so is not here"
ref = ol.Reference()
rng = np.random.default_rng(20261003 + seed)
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(2); os.dup2(devnull, 2)
try:
    mod.kat_elliptic(ref, rng)
    mod.kat_geodesic(ref, rng)
    mod.kat_kerr(ref, rng)
    mod.kat_polar(ref, rng)
    mod.kat_vectors(ref)
    mod.kat_azimuth(ref)
    mod.kat_boundary()
    mod.kat_kerr_newman()
    mod.kat_raytrace(ref, rng)
    mod.kat_init_src()
    mod.kat_disk_edge()
    mod.kat_disk_model()
finally:
    os.dup2(saved, 2)
env = dict(os.environ, SIM5_GOLDEN_DIR=out)
tests = ["tests/test_gpu_kat.py::test_elliptic", "tests/test_gpu_kat.py::test_geodesic_init_inf_records", "tests/test_gpu_kat.py::test_kerr",
         "tests/test_gpu_kat.py::test_vectors", "tests/test_gpu_kat.py::test_polarization_and_blackbody", "tests/test_gpu_kat.py::test_geodesic_chain_records_both_arithmetics",
         "tests/test_gpu_kat.py::test_azimuth_integrals", "tests/test_gpu_kat.py::test_position_azm_and_timedelay", "tests/test_gpu_kat.py::test_boundary_prototypes",
         "tests/test_gpu_kat.py::test_kerr_newman_prototypes", "tests/test_gpu_kat.py::test_geodesic_init_src_records", "tests/test_gpu_kat.py::test_disk_flux_inner_edge_band", "tests/test_gpu_kat.py::test_disk_model_rest",
         "tests/test_gpu_raytrace.py::test_prepare_and_single_step", "tests/test_gpu_raytrace.py::test_step_sequences_follow_reference"]
rc = subprocess.call([sys.executable, "-m", "pytest", "-q", "-s"] + tests, cwd=root, env=env)
shutil.rmtree(out, ignore_errors=True)
print("fuzz_kat: scale %d seed %d: pytest rc %d" % (scale, seed, rc))
sys.exit(rc)
