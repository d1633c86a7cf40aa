# fast / strict image kernels against the CPU checker over a sweep of spins and inclinations (256^2 each)
import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
from gpuutil import deg2rad, oraclelib as ol
rng=np.random.default_rng(11)
cfgs=[(0.0,1.0),(0.0,89.0),(0.999999,89.9),(0.999999,0.5),(1e-5,45.0),(0.5,5.0),(0.998,85.0),(0.9999,60.0)]
cfgs+=[(float(rng.uniform(0,0.9999)), float(rng.uniform(1,89))) for _ in range(24)]
n=256; tot_f=tot_s=0
for a,inc in cfgs:
    c=ol.cpu_disk_image("port", n, n, a, inc, nthreads=8, full=True)
    for strict in (False, True):
        d=capi.image_desc(n,n,a,deg2rad(inc),strict=strict)
        o=capi.disk_image(d, full=True)
        flips=int((o["cls"]!=c["cls"]).sum())
        hit=(c["cls"]==2)|(c["cls"]==4); hit&=(o["cls"]==c["cls"])
        er=np.max(np.abs(o["r"][hit]/c["r"][hit]-1)) if hit.any() else 0
        eg=np.max(np.abs(o["g"][hit]/c["g"][hit]-1)) if hit.any() else 0
        fl=np.max(np.abs(o["flux"][hit]-c["flux"][hit])/np.maximum(np.abs(c["flux"][hit]),1e-9*c["flux"].max())) if hit.any() else 0
        if strict: tot_s+=flips
        else: tot_f+=flips
        if flips or er>1e-9 or eg>1e-9 or fl>1e-6: print("a=%.6f inc=%.2f %s flips %d r %.2e g %.2e flux %.2e"%(a,inc,"strict" if strict else "fast",flips,er,eg,fl))
print("configs",len(cfgs),"total flips fast",tot_f,"strict",tot_s)
