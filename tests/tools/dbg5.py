import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
import test_gpu_raytrace as T
n=1024; N=n*n
dd=T.torus_desc(capi,n,0.9,70.0,r0=100.0,precision=1.0,max_steps=100000)
sb=capi.DeviceBuffer(N*40); steps=capi.DeviceBuffer(N*4)
for rep in range(3):
    steps.zero()
    for k in range(rep+1):
        capi.torus_image_device(dd, sb.ptr, aux={"steps":steps.ptr})
    capi.synchronize()
    s=steps.to_numpy(np.int32,(N,))
    print("launches back-to-back:", rep+1, "mean", s.mean(), "zeros", (s==0).sum(), np.histogram(s, bins=[0,1,2,10,50,100,300,600,2000,100000])[0].tolist())
