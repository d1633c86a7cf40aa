import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
def timeit(fn, reps=5, warm=1):
    for _ in range(warm): fn()
    capi.synchronize(); e0=capi.Event(); e1=capi.Event(); e0.record()
    for _ in range(reps): fn()
    e1.record(); return e0.elapsed_ms(e1)/reps
# C3 polarized 2048^2 a=0.9
n=2048; d=capi.image_desc(n,n,0.9,70/180*math.pi,pol_degree=0.1)
st=capi.DeviceBuffer(3*n*n*8); ch=capi.DeviceBuffer(n*n*8)
ms=timeit(lambda: capi.disk_image_polarized_device(d, st.ptr, ch.ptr))
print("C3 polarized 2048^2 fast: %.3f ms  %.3e rays/s"%(ms, n*n/ms*1e3))
d.flags=1
ms=timeit(lambda: capi.disk_image_polarized_device(d, st.ptr, ch.ptr))
print("C3 polarized 2048^2 strict: %.3f ms  %.3e rays/s"%(ms, n*n/ms*1e3))
# C2 1024^2
n=1024; d=capi.image_desc(n,n,0.998,70/180*math.pi); f=capi.DeviceBuffer(n*n*4); g=capi.DeviceBuffer(n*n*4)
ms=timeit(lambda: capi.disk_image_device(d,f.ptr,g.ptr), reps=20)
print("C2 1024^2 fast: %.3f ms  %.3e rays/s"%(ms, n*n/ms*1e3))
# C4 torus 1024^2 Verlet
import test_gpu_raytrace as T
n=1024
for prec in (1.0, 0.01):
    dd=T.torus_desc(capi,n,0.9,70.0,r0=100.0,precision=prec,max_steps=100000)
    N=n*n; sb=capi.DeviceBuffer(N*40); steps=capi.DeviceBuffer(N*4)
    ms=timeit(lambda: capi.torus_image_device(dd, sb.ptr, aux={"steps":steps.ptr}), reps=2, warm=1)
    s=steps.to_numpy(np.int32,(N,))
    tot=int(s.sum())
    print("C4 torus 1024^2 precision %g: %.1f ms  %.3e rays/s  mean steps %.1f  %.3e steps/s  W_step frac=%.4f"%(prec, ms, N/ms*1e3, s.mean(), tot/ms*1e3, tot*750/ms*1e3/78.6e12))
    print("   hist", np.histogram(s, bins=[0,1,2,10,50,100,300,600,2000,100000])[0].tolist())
# spectrum: 1024^2 pixels x 128 energies
n=1024; d=capi.image_desc(n,n,0.998,70/180*math.pi)
E=10.0**np.linspace(-1,1.5,128)
import ctypes as C
capi._lib.sim5gpu_disk_spectrum_workspace.restype = capi.SZ
ws_bytes = capi._lib.sim5gpu_disk_spectrum_workspace(C.byref(d), capi.I(E.size))
dE=capi.DeviceBuffer(E.nbytes); dS=capi.DeviceBuffer(E.nbytes); ws=capi.DeviceBuffer(ws_bytes); dE.from_numpy(E)
fn=lambda: capi._check(capi._lib.sim5gpu_disk_spectrum(C.byref(d), capi.I(E.size), capi.VP(dE.ptr), capi.D(1.7), capi.I(1), capi.VP(dS.ptr), capi.VP(ws.ptr), capi.VP(0)),"spec")
ms=timeit(fn, reps=5)
print("spectrum 1024^2 x 128 energies: %.3f ms  %.3e rays/s  %.3e (ray,energy) pairs/s"%(ms, n*n/ms*1e3, n*n*128/ms*1e3))
# surface search (thick disk H(R) = 0.25 (R - 2), 256-point table), 1024^2 rays
import ctypes as C
n=1024; a=0.9; inc=70/180*math.pi
rmax=20.0
ax=((np.arange(n)+.5)/n-.5)*2*rmax
al,be=np.meshgrid(ax,ax); al=al.ravel().copy(); be=be.ravel().copy()
tR=np.linspace(2.0,60.0,256); tH=0.25*(tR-2.0)
N=al.size
b={k:capi.DeviceBuffer(v.nbytes) for k,v in (("tR",tR),("tH",tH),("al",al),("be",be))}
for k,v in (("tR",tR),("tH",tH),("al",al),("be",be)): b[k].from_numpy(v)
o={k:capi.DeviceBuffer(N*s) for k,s in (("P",8),("r",8),("m",8),("k",32),("st",4))}
for strict in (0,1):
    fn=lambda: capi._check(capi._lib.sim5gpu_disk_surface_rays(capi.D(a),capi.D(inc),capi.I(tR.size),capi.VP(b["tR"].ptr),capi.VP(b["tH"].ptr),capi.SZ(N),capi.VP(b["al"].ptr),capi.VP(b["be"].ptr),capi.VP(o["P"].ptr),capi.VP(o["r"].ptr),capi.VP(o["m"].ptr),capi.VP(o["k"].ptr),capi.VP(o["st"].ptr),capi.I(strict),capi.VP(0)),"surf")
    ms=timeit(fn,reps=3)
    st=o["st"].to_numpy(np.int32,(N,))
    print("surface search 1024^2 %s: %.3f ms  %.3e rays/s  hits %d"%("strict" if strict else "fast",ms,N/ms*1e3,int((st==1).sum())))
