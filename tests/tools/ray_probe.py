"""One ray from infinity through geodesic_init_inf on the GPU, strict record and fast record side by side
(python tests/tools/ray_probe.py a incl_deg alpha beta)."""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sim5_amd.capi as capi
a, inc, al, be = (float(v) for v in sys.argv[1:5])
for fast in (False, True):
    g, err, ok, ch = capi.geodesic_init_inf_chain(math.radians(inc), a, [al], [be], fast=fast)
    print("fast" if fast else "strict", "ok", ok[0], "err", err[0], "l %.17g q %.17g m2p %.17g m2m %.17g" % (g["l"][0], g["q"][0], g["m2p"][0], g["m2m"][0]))
