"""Central column / row of an odd-sized image: classes and error codes of the reference, the strict and the fast variant side by side
(run on the GPU box: python tests/tools/column_diag.py a incl_deg nx ny)."""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib as ol
import sim5_amd.capi as capi
from gpuutil import deg2rad
a, inc, nx, ny = float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ref = ol.cpu_disk_image("reference", nx, ny, a, inc, nthreads=8, full=True)
st = capi.disk_image(capi.image_desc(nx, ny, a, deg2rad(inc), strict=True), full=True)
fa = capi.disk_image(capi.image_desc(nx, ny, a, deg2rad(inc), strict=False), full=True)
print("keys", sorted(fa.keys()))
if nx % 2:
    c = nx // 2
    for iy in range(ny):
        flag = "" if ref["cls"][iy, c] == fa["cls"][iy, c] else "  <-- fast differs"
        print("col iy=%3d ref %d strict %d fast %d   gtype %s %s %s%s" % (iy, ref["cls"][iy, c], st["cls"][iy, c], fa["cls"][iy, c],
              ref["gtype"][iy, c], st["gtype"][iy, c], fa["gtype"][iy, c], flag))
if ny % 2:
    r = ny // 2
    for ix in range(nx):
        if ref["cls"][r, ix] != fa["cls"][r, ix] or ref["cls"][r, ix] != st["cls"][r, ix]:
            print("row ix=%3d ref %d strict %d fast %d" % (ix, ref["cls"][r, ix], st["cls"][r, ix], fa["cls"][r, ix]))
