"""Where along ONE ray do the fast and the strict march part?  The job is run with max_steps = 1, 2, ... in both variants and
the ray's state after each count is compared (python tests/tools/torus_diverge.py a incl_deg n r0 ray_index [precision])."""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sim5_amd.capi as capi
from test_gpu_raytrace import torus_desc, run_torus
a, inc, n, r0, ray = float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
prec = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0
last = None
for ms in list(range(1, 400)):
    out = []
    for strict in (True, False):
        d = torus_desc(capi, n, a, inc, r0=r0, precision=prec, max_steps=ms)
        if strict:
            d.img.flags = 1
        S, steps, xe, ce, me, ke = run_torus(capi, d, full=True)
        out.append((steps[ray], xe[ray].copy(), ke[ray].copy()))
    (s0, x0, k0), (s1, x1, k1) = out
    dx = np.abs(x1 - x0) / np.maximum(np.abs(x0), 1e-2); dk = np.abs(k1 - k0) / np.maximum(np.abs(k0), 1e-4)
    print("steps %4d/%4d  r %.9f m %+.12f  1-|m| %.3e  dx %s dk %s" % (s0, s1, x0[1], x0[2], 1 - abs(x0[2]),
          " ".join("%.1e" % v for v in dx), " ".join("%.1e" % v for v in dk)))
    if s0 < ms:
        break
