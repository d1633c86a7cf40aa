import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
n=4096
o=capi.disk_image(capi.image_desc(n,n,0.998,70/180*math.pi), full=True)
t=o["gtype"].astype(int)[:n//2]            # upper half: the lanes of the mirror kernel
print("depth histogram", {int(k):int(v) for k,v in zip(*np.unique(t,return_counts=True))})
p=t.reshape(n//2//4,4,n//16,16).transpose(0,2,1,3).reshape(-1,64)
use=p>=0
rungs=np.where(use,p+1,0)
mx=rungs.max(axis=1)
act=mx>0
print("waves using the ladder %.3f"%act.mean())
print("rung-loop lane use %.3f"%(rungs[act].sum()/(64*mx[act].sum())))
print("lanes using ladder in active waves %.3f"%use[act].mean())
print("waves with mixed depth %.3f"%((np.where(use,p,99).min(axis=1)!=mx-1)[act].mean()))
