import sys, math, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
import oraclelib as ol
for (a, inc, nx, ny, order, rmax) in [(0.9999, 53.25, 47, 19, 2, 0.0), (0.9, 59.82, 287, 46, 2, 0.0), (0.791869, 55.26, 113, 296, 1, 0.0)]:
    mk = lambda strict: capi.disk_image(capi.image_desc(nx, ny, a, math.radians(inc), max_order=order, rmax=rmax, strict=strict), full=True)
    f, s = mk(False), mk(True)
    c = ol.cpu_disk_image("port", nx, ny, a, inc, nthreads=8, full=True) if order == 2 else None
    d = np.argwhere(f["cls"] != s["cls"])
    print("case", a, inc, nx, ny, "fast!=strict:", len(d))
    for (y, x) in d[:6]:
        print("   px", y, x, "fast cls", f["cls"][y, x], "r", f["r"][y, x], "| strict cls", s["cls"][y, x], "r", s["r"][y, x], "gtype", f["gtype"][y, x], s["gtype"][y, x])
    if c is not None:
        d = np.argwhere(c["cls"] != s["cls"])
        print("   strict!=oracle:", len(d))
        for (y, x) in d[:6]:
            print("   px", y, x, "oracle cls", c["cls"][y, x], "r", c["r"][y, x], "| strict cls", s["cls"][y, x], "r", s["r"][y, x], "gtype", c["gtype"][y, x], s["gtype"][y, x])
