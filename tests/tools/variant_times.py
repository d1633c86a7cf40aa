import sys, math
sys.path.insert(0,'.')
import sim5_amd.capi as capi
n=4096
for strict in (False, True):
    d=capi.image_desc(n,n,0.998,math.radians(70),strict=strict)
    f=capi.DeviceBuffer(n*n*4); g=capi.DeviceBuffer(n*n*4)
    capi.disk_image_device(d,f.ptr,g.ptr); capi.synchronize()
    e0=capi.Event(); e1=capi.Event(); e0.record()
    for _ in range(10): capi.disk_image_device(d,f.ptr,g.ptr)
    e1.record(); ms=e0.elapsed_ms(e1)/10
    print("strict" if strict else "fast", "%.3f ms %.3e rays/s"%(ms, n*n/ms*1e3))
