"""One case of fuzz_images.py (seed, case): its central column / row with the classes of the live reference, the CPU port, the
strict and the fast variant side by side (python tests/tools/fuzz_column_case.py seed case)."""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib as ol
import sim5_amd.capi as capi
from gpuutil import deg2rad
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(want + 1):
    a = float(rng.choice([0.0, 1e-5, 0.3, 0.7, 0.9, 0.998, 0.9999, rng.uniform(0, 0.999)]))
    inc = float(rng.uniform(3.0, 87.0))
    nx, ny = int(rng.integers(17, 300)), int(rng.integers(2, 300))
    order = int(rng.choice([1, 2]))
    rmax = float(rng.choice([0.0, rng.uniform(3.0, 60.0)]))
print("case %d: a=%.17g inc=%.17g %dx%d order %d rmax %g" % (want, a, inc, nx, ny, order, rmax))
ref = ol.cpu_disk_image("reference", nx, ny, a, inc, nthreads=8, full=True) if ol.have_reference() else None
port = ol.cpu_disk_image("port", nx, ny, a, inc, nthreads=8, full=True)
st = capi.disk_image(capi.image_desc(nx, ny, a, deg2rad(inc), strict=True), full=True)
fa = capi.disk_image(capi.image_desc(nx, ny, a, deg2rad(inc), strict=False), full=True)
d = (port["cls"] != st["cls"]) | (fa["cls"] != st["cls"]) | ((ref["cls"] != st["cls"]) if ref is not None else False)
for iy, ix in zip(*np.nonzero(d)):
    print("pixel (%d,%d): reference %s port %d strict %d fast %d   r ref %s port %.9g strict %.9g" % (iy, ix, ref["cls"][iy, ix] if ref is not None else "-", port["cls"][iy, ix],
          st["cls"][iy, ix], fa["cls"][iy, ix], ("%.9g" % ref["r"][iy, ix]) if ref is not None else "-", port["r"][iy, ix], st["r"][iy, ix]))
print("differing pixels: %d" % int(d.sum()))
