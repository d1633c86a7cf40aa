import sys, math, numpy as np, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
import oraclelib as ol
n=4096
f = capi.disk_image(capi.image_desc(n,n,0.998,70/180*math.pi), full=True)
import os, pickle
cp='/tmp/cpu4096.npz'
if not os.path.exists(cp):
    c = ol.cpu_disk_image("port", n,n,0.998,70.0, nthreads=64, full=True); np.savez(cp, **{k:v for k,v in c.items() if k in("r","g","flux","cls")})
c=np.load(cp)
flips = int((f["cls"]!=c["cls"]).sum())
out=["flips %d"%flips]
for k in ("r","g","flux"):
    a,b=f[k],c[k]; m=~np.isnan(b)&(np.abs(b)>0)&~np.isnan(a)
    e=np.abs(a[m]-b[m])/np.abs(b[m])
    out.append("%s max %.2e  >1e-10: %d  >1e-12: %d"%(k,e.max(),(e>1e-10).sum(),(e>1e-12).sum()))
d=capi.image_desc(n,n,0.998,70/180*math.pi)
bf=capi.DeviceBuffer(n*n*4); bg=capi.DeviceBuffer(n*n*4)
for _ in range(200 if "--warm" in sys.argv else 3): capi.disk_image_device(d,bf.ptr,bg.ptr)   # --warm: ~80 ms of launches first (working clock)
capi.synchronize(); e0=capi.Event(); e1=capi.Event(); e0.record()
for _ in range(50): capi.disk_image_device(d,bf.ptr,bg.ptr)
e1.record(); ms=e0.elapsed_ms(e1)/50
print(" | ".join(out), "| %.3f ms  %.3e rays/s"%(ms, n*n/ms*1e3))
