import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
n=4096; a=0.998
o=capi.disk_image(capi.image_desc(n,n,a,70/180*math.pi), full=True)
r=o["r"]; hit=np.isin(o["cls"],(2,4)) if o["cls"].max()>2 else (o["flux"]>=0)&np.isfinite(r)
rms=float(capi.r_ms([a])[0]); x0=math.sqrt(rms); x=np.sqrt(np.where(np.isfinite(r),r,1e9))
hit=np.isfinite(r)
near=hit&(x-x0<=2e-4)&(r>rms); far=hit&(x0/x<=x0/16)
zero=hit&(o["flux"]==0)
def tiles(m): return m.reshape(n//4,4,n//16,16).any(axis=(1,3)).mean()
print("hit frac %.4f near-edge px %d far px %d zero-flux hit px %d | tiles with near %.5f far %.5f zero %.5f"%(hit.mean(),near.sum(),far.sum(),zero.sum(),tiles(near),tiles(far),tiles(zero)))
print("r max %.2f"%np.nanmax(r))
