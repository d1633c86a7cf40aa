import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
import test_gpu_raytrace as T
for n in (24, 96, 256, 512, 1024):
    dd=T.torus_desc(capi,n,0.9,70.0,r0=100.0,precision=1.0,max_steps=100000)
    S, steps, xe, ce, me = T.run_torus(capi, dd)
    print(n, "mean steps %.1f  zeros %d  hist"%(steps.mean(), (steps==0).sum()), np.histogram(steps, bins=[0,1,2,10,50,100,300,600,2000,100000])[0].tolist(), "r_end<2: %d  r_end>100: %d  other: %d"%((xe[:,1]<2).sum(), (xe[:,1]>100).sum(), ((xe[:,1]>=2)&(xe[:,1]<=100)).sum()), "maxerr>1e-2:", int((me>1e-2).sum()))
    if n==256:
        s2=steps.reshape(n,n); print(s2[::32, ::32])
