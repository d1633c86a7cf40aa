import sys, math
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import sim5_amd.capi as capi
from test_gpu_raytrace import torus_desc, run_torus
g=np.load('/root/repo/tests/golden/torus_c4.npz')
n=1024;a=0.9;inc=70.0
sel=g["thin_iy"].astype(np.int64)*n+g["thin_ix"]
for strict in (0,1):
    d=torus_desc(capi,n,a,inc)
    if strict: d.img.flags=1
    S,steps,xe,ce,me,ke=run_torus(capi,d,full=True)
    same=steps[sel]==g["thin_steps"]
    print("strict" if strict else "fast", "same", same.sum(), "of", same.size, "diffs", np.unique((steps[sel]-g["thin_steps"])[~same]))
    er=np.abs(xe[sel][:,1]-g["thin_x_end"][:,1])/g["thin_x_end"][:,1]
    ei=np.abs(S[sel][:,0]-g["thin_I"])/np.maximum(g["thin_I"],1e-6*g["thin_I"].max())
    for nm,e in (("r_end",er),("I",ei)):
        e=e[same]
        print(nm, "q50 %.1e q99 %.1e q999 %.1e max %.1e  n>1e-6: %d n>1e-9: %d"%(np.quantile(e,.5),np.quantile(e,.99),np.quantile(e,.999),e.max(),(e>1e-6).sum(),(e>1e-9).sum()))
    bad=np.nonzero(same&(er>1e-7))[0]
    print("bad rays: steps", g["thin_steps"][bad][:10], "ix", g["thin_ix"][bad][:10], "iy", g["thin_iy"][bad][:10], "carter", g["thin_carter"][bad][:10])
