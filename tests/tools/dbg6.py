import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
import test_gpu_raytrace as T
n=1024; N=n*n
def hist(s): return "mean %.1f zeros %d"%(s.mean(), (s==0).sum())
dd=T.torus_desc(capi,n,0.9,70.0,r0=100.0,precision=1.0,max_steps=100000)
# A: fresh buffers, no zero, single launch
sb=capi.DeviceBuffer(N*40); steps=capi.DeviceBuffer(N*4)
capi.torus_image_device(dd, sb.ptr, aux={"steps":steps.ptr}); capi.synchronize()
print("A fresh single:", hist(steps.to_numpy(np.int32,(N,))))
# B: events around
e0=capi.Event(); e1=capi.Event(); e0.record()
capi.torus_image_device(dd, sb.ptr, aux={"steps":steps.ptr})
e1.record(); ms=e0.elapsed_ms(e1)
print("B with events: %.1f ms"%ms, hist(steps.to_numpy(np.int32,(N,))))
# C: after an image kernel
d=capi.image_desc(n,n,0.998,70/180*math.pi); f=capi.DeviceBuffer(n*n*4); g=capi.DeviceBuffer(n*n*4)
capi.disk_image_device(d,f.ptr,g.ptr)
steps.zero()
e0.record(); capi.torus_image_device(dd, sb.ptr, aux={"steps":steps.ptr}); e1.record(); ms=e0.elapsed_ms(e1)
print("C after image kernel: %.1f ms"%ms, hist(steps.to_numpy(np.int32,(N,))))
# D: polarized before
n2=2048; d2=capi.image_desc(n2,n2,0.9,70/180*math.pi,pol_degree=0.1); st=capi.DeviceBuffer(3*n2*n2*8)
capi.disk_image_polarized_device(d2, st.ptr, None)
steps.zero()
e0.record(); capi.torus_image_device(dd, sb.ptr, aux={"steps":steps.ptr}); e1.record(); ms=e0.elapsed_ms(e1)
print("D after polarized kernel: %.1f ms"%ms, hist(steps.to_numpy(np.int32,(N,))))
