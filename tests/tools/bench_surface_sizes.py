# ON THE GPU BOX: surface-search time against the number of rays (T = a + b N: what the iteration's tail costs)
import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
a=0.9; inc=70/180*math.pi; rmax=20.0
tR=np.linspace(2.0,60.0,256); tH=0.25*(tR-2.0)
rows=[]
for n in (256, 512, 1024, 2048):
    ax=((np.arange(n)+.5)/n-.5)*2*rmax
    al,be=np.meshgrid(ax,ax); al=al.ravel().copy(); be=be.ravel().copy()
    N=al.size
    b={k:capi.DeviceBuffer(v.nbytes) for k,v in (("tR",tR),("tH",tH),("al",al),("be",be))}
    for k,v in (("tR",tR),("tH",tH),("al",al),("be",be)): b[k].from_numpy(v)
    o={k:capi.DeviceBuffer(N*s) for k,s in (("P",8),("r",8),("m",8),("k",32),("st",4))}
    fn=lambda: capi._check(capi._lib.sim5gpu_disk_surface_rays(capi.D(a),capi.D(inc),capi.I(tR.size),capi.VP(b["tR"].ptr),capi.VP(b["tH"].ptr),capi.SZ(N),capi.VP(b["al"].ptr),capi.VP(b["be"].ptr),capi.VP(o["P"].ptr),capi.VP(o["r"].ptr),capi.VP(o["m"].ptr),capi.VP(o["k"].ptr),capi.VP(o["st"].ptr),capi.I(0),capi.VP(0)),"surf")
    for _ in range(20): fn()
    capi.synchronize(); e0=capi.Event(); e1=capi.Event(); e0.record()
    for _ in range(5): fn()
    e1.record(); ms=e0.elapsed_ms(e1)/5
    rows.append((N,ms)); print("%5d^2  %.3f ms  %.3e rays/s"%(n,ms,N/ms*1e3),flush=True)
A=np.array([[1.0,r[0]] for r in rows]); y=np.array([r[1] for r in rows])
(c0,c1),*_=np.linalg.lstsq(A,y,rcond=None)
print("fit T = a + b N: a = %.3f ms, 1/b = %.3e rays/s"%(c0,1e3/c1))
