import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
import oraclelib as ol
def run(n,a,inc,strict):
    return capi.disk_image(capi.image_desc(n,n,a,inc/180*math.pi,strict=strict), full=True)
f = run(4096,0.998,70.0,False); s = run(4096,0.998,70.0,True)
c = ol.cpu_disk_image("port", 4096,4096,0.998,70.0, nthreads=64, full=True)
for name,(A,B) in {"fast-strict":(f,s),"fast-cpu":(f,c),"strict-cpu":(s,c)}.items():
    for k in ("r","g","flux"):
        a,b = A[k],B[k]
        m = ~np.isnan(b) & (np.abs(b)>0)
        e = np.zeros_like(a); e[m] = np.abs(a[m]-b[m])/np.abs(b[m])
        iy,ix = np.unravel_index(np.argmax(e), e.shape)
        print(name,k,"max rel %.3e at (%d,%d) val %.17g vs %.17g r=%.6g cls=%d type=%d"%(e.max(),iy,ix,a[iy,ix],b[iy,ix],B["r"][iy,ix],B["cls"][iy,ix],B["gtype"][iy,ix]), " p99.99=%.2e median=%.2e"%(np.quantile(e[m],0.9999), np.median(e[m])))
# histogram of fast-cpu r errors
a,b=f["r"],c["r"]; m=~np.isnan(b)
e=np.abs(a[m]-b[m])/np.abs(b[m])
for t in (1e-14,1e-13,1e-12,1e-11,1e-10,1e-9,1e-8,1e-7):
    print("r err >",t, int((e>t).sum()))
