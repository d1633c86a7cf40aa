# ON THE GPU BOX: the C4 job in the STRICT variant (kernel time by HIP events, checksum)
import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
import test_gpu_raytrace as T
n=1024; N=n*n
dd=T.torus_desc(capi,n,0.9,70.0,r0=100.0,precision=1.0,max_steps=100000)
dd.img.flags = 1
sb=capi.DeviceBuffer(N*40); steps=capi.DeviceBuffer(N*4)
capi.torus_image_device(dd, sb.ptr, aux={"steps":steps.ptr}); capi.synchronize()
e0=capi.Event(); e1=capi.Event(); e0.record()
for _ in range(2): capi.torus_image_device(dd, sb.ptr, aux={"steps":steps.ptr})
e1.record(); ms=e0.elapsed_ms(e1)/2
s=steps.to_numpy(np.int32,(N,)); S=sb.to_numpy(np.float64,(N,5))
print("strict %.1f ms  mean steps %.1f  sumI %.12e"%(ms, s.mean(), S[:,0].sum()))
