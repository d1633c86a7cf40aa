import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
n=int(sys.argv[1]) if len(sys.argv)>1 else 1024; a=0.9; inc=70/180*math.pi; rmax=20.0
ax=((np.arange(n)+.5)/n-.5)*2*rmax
al,be=np.meshgrid(ax,ax); al=al.ravel().copy(); be=be.ravel().copy()
tR=np.linspace(2.0,60.0,256); tH=0.25*(tR-2.0)
N=al.size
b={k:capi.DeviceBuffer(v.nbytes) for k,v in (("tR",tR),("tH",tH),("al",al),("be",be))}
for k,v in (("tR",tR),("tH",tH),("al",al),("be",be)): b[k].from_numpy(v)
o={k:capi.DeviceBuffer(N*s) for k,s in (("P",8),("r",8),("m",8),("k",32),("st",4))}
out=[]
for strict in (0,1):
    fn=lambda: capi._check(capi._lib.sim5gpu_disk_surface_rays(capi.D(a),capi.D(inc),capi.I(tR.size),capi.VP(b["tR"].ptr),capi.VP(b["tH"].ptr),capi.SZ(N),capi.VP(b["al"].ptr),capi.VP(b["be"].ptr),capi.VP(o["P"].ptr),capi.VP(o["r"].ptr),capi.VP(o["m"].ptr),capi.VP(o["k"].ptr),capi.VP(o["st"].ptr),capi.I(strict),capi.VP(0)),"surf")
    fn(); capi.synchronize(); e0=capi.Event(); e1=capi.Event(); e0.record()
    for _ in range(3): fn()
    e1.record(); ms=e0.elapsed_ms(e1)/3
    st=o["st"].to_numpy(np.int32,(N,)); r=o["r"].to_numpy(np.float64,(N,))
    out.append("%s %.3f ms hits %d sum_r %.10e"%("strict" if strict else "fast",ms,int((st==1).sum()),float(r[st==1].sum())))
print("n=%d: "%n + " | ".join(out))
