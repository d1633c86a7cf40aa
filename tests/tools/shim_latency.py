"""ON THE GPU BOX: where a scalar-API round trip goes -- a trivial n = 1 batch call (launch + completion only) against the
chain call of geodesic_init_inf (the same plus one ray's dependent FP64 chain in two lanes), microseconds per call."""
import sys, time, math, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import sim5_amd.capi as capi
lib = capi._lib
a = np.array([0.9]); out = np.zeros(1)
def t(fn, n=3000):
    for _ in range(200): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
triv = lambda: lib.sim5gpu_r_bh(capi.SZ(1), capi._p(a), capi._p(out))
inc = np.array([math.radians(70.0)]); sp = np.array([0.998]); al = np.array([3.0]); be = np.array([4.0])
g = np.zeros(1, dtype=capi.GEODESIC_DTYPE); err = np.zeros(1, np.int32); ok = np.zeros(1, np.int32)
class Chain(C.Structure):
    _fields_ = [("P", C.c_double * 2), ("r", C.c_double * 2), ("g", C.c_double * 2), ("flux", C.c_double * 2), ("a", C.c_double), ("l", C.c_double),
                ("have_r", C.c_int * 2), ("valid", C.c_int), ("flux_valid", C.c_int)]
ch = Chain()
capi.disk_nt_setup(10.0, 0.998, 0.1, 0.1)
chain = lambda: lib.sim5gpu_geodesic_init_inf_chain(capi.SZ(1), capi._p(inc), capi._p(sp), capi._p(al), capi._p(be), capi._p(g), capi._p(err), capi._p(ok), C.byref(ch))
chain_fast = lambda: lib.sim5gpu_geodesic_init_inf_chain_fast(capi.SZ(1), capi._p(inc), capi._p(sp), capi._p(al), capi._p(be), capi._p(g), capi._p(err), capi._p(ok), C.byref(ch))
plain = lambda: lib.sim5gpu_geodesic_init_inf(capi.SZ(1), capi._p(inc), capi._p(sp), capi._p(al), capi._p(be), capi._p(g), capi._p(err), capi._p(ok))
print("trivial call %.1f us | geodesic_init_inf %.1f us | chain (init_inf + crossings + radii + g + flux) %.1f us | r0 %.15g" % (t(triv), t(plain), t(chain), ch.r[0]))
print("the chain in the fast arithmetic %.1f us | r0 %.15g flux0 %.15g" % (t(chain_fast), ch.r[0], ch.flux[0]))
