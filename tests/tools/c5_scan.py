import sys, math, numpy as np
sys.path.insert(0,'.')
import sim5_amd.capi as capi
n=8192
f=capi.DeviceBuffer(n*n*4); g=capi.DeviceBuffer(n*n*4)
tot=0
for inc in (10,20,30,40,50,60,70,80):
    d=capi.image_desc(n,n,0.998,math.radians(inc))
    capi.disk_image_device(d,f.ptr,g.ptr); capi.synchronize()
    e0=capi.Event(); e1=capi.Event(); e0.record()
    for _ in range(3): capi.disk_image_device(d,f.ptr,g.ptr)
    e1.record(); ms=e0.elapsed_ms(e1)/3; tot+=ms
    print("i=%d: %.3f ms %.3e rays/s"%(inc, ms, n*n/ms*1e3))
print("C5 (8 inclinations, 8192^2 each) on ONE GPU: %.1f ms total, %.3e rays/s"%(tot, 8*n*n/tot*1e3))
