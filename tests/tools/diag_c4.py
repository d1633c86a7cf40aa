"""ON THE GPU BOX: the C4 job against the golden rays of the reference (tests/golden/torus_c4.npz): step counts, worst
relative differences of the end state and of Stokes I per variant, and the job's time.  SIM5GPU_LIB selects the library."""
import sys, math, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sim5_amd.capi as capi
from test_gpu_raytrace import torus_desc, run_torus
g = np.load(os.path.join(ROOT, "tests/golden/torus_c4.npz"))
n = 1024; a = 0.9; inc = 70.0
sel = g["thin_iy"].astype(np.int64) * n + g["thin_ix"]
variants = (0, 1) if "--both" in sys.argv else (0,)
for strict in variants:
    d = torus_desc(capi, n, a, inc)
    if strict: d.img.flags = 1
    S, steps, xe, ce, me, ke = run_torus(capi, d, full=True)
    e0 = capi.Event(); e1 = capi.Event()
    st = capi.DeviceBuffer(n * n * 40)
    capi.torus_image_device(d, st.ptr); capi.synchronize()
    e0.record()
    for _ in range(3): capi.torus_image_device(d, st.ptr)
    e1.record(); ms = e0.elapsed_ms(e1) / 3
    same = steps[sel] == g["thin_steps"]
    out = ["strict" if strict else "fast", "%.2f ms" % ms, "same steps %d/%d" % (same.sum(), same.size)]
    ref_x = g["thin_x_end"]; ref_k = g["thin_k_end"]
    def rel(got, want, floor): return np.abs(got - want) / np.maximum(np.abs(want), floor)
    errs = {"t": rel(xe[sel][:, 0], ref_x[:, 0], 1.0), "r": rel(xe[sel][:, 1], ref_x[:, 1], 0.0),
            "m": rel(xe[sel][:, 2], ref_x[:, 2], 1e-2), "phi": rel(xe[sel][:, 3], ref_x[:, 3], 1.0),
            "k1": rel(ke[sel][:, 1], ref_k[:, 1], 1e-2), "k2": rel(ke[sel][:, 2], ref_k[:, 2], 1e-4),
            "k3": rel(ke[sel][:, 3], ref_k[:, 3], 1e-4),
            "I": rel(S[sel][:, 0], g["thin_I"], 1e-6 * g["thin_I"].max())}
    tot = np.zeros(sel.size)
    for k, e in errs.items():
        e = np.where(same, e, 0.0); tot = np.maximum(tot, e)
        out.append("%s %.1e" % (k, e.max()))
    out.append("rays>1e-6: %d  >1e-7: %d  median %.1e" % ((tot > 1e-6).sum(), (tot > 1e-7).sum(), np.median(tot)))
    print("  ".join(out))
