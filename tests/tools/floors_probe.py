import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import sim5_amd.capi as capi
from gpuutil import rel_err
g=np.load('/root/repo/tests/golden/kat_geodesic.npz')
inp=g["inp"]
rec,err,ok=capi.geodesic_init_inf(inp[:,0],inp[:,1],inp[:,2],inp[:,3])
ref=np.frombuffer(g["dump"].tobytes(),dtype=capi.GEODESIC_DTYPE)
good=ok==1
for f in ("rp","Tip","r1","r2","r3","r4"):
    for fl in (1e-3,1e-6,1e-9,0.0):
        try: e=rel_err(rec[f][good],ref[f][good],fl)
        except AssertionError as ex: e=str(ex)
        print(f,fl,e)
    a=np.asarray(rec[f][good],float); b=np.asarray(ref[f][good],float)
    d=np.abs(a-b); print("   max abs diff %.3e"%np.nanmax(d), "min |ref| %.3e"%np.nanmin(np.abs(b[np.isfinite(b)])))
g=np.load('/root/repo/tests/golden/kat_azimuth.npz')
for name in capi.INTEGRALS:
    args=g["in_"+name]; got=capi.integral(name,*[args[:,k] for k in range(args.shape[1])]); ref=g["out_"+name]
    print(name, "floor1e-3: %.2e  floor1e-9: %.2e  maxabs %.2e  typical |ref| %.2e"%(rel_err(got,ref,1e-3),rel_err(got,ref,1e-9),np.nanmax(np.abs(got-ref)),np.nanmedian(np.abs(ref))))
