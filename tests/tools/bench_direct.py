import sys, math, numpy as np
sys.path.insert(0,'.')
import sim5_amd.capi as capi, time
n=4096
bf=capi.DeviceBuffer(n*n*4); bg=capi.DeviceBuffer(n*n*4)
for direct in (False, True, False, True):
    d=capi.image_desc(n,n,0.998,70/180*math.pi, direct=direct)
    t0=time.time()
    while time.time()-t0<0.4: capi.disk_image_device(d,bf.ptr,bg.ptr)
    capi.synchronize(); e0=capi.Event(); e1=capi.Event(); e0.record()
    for _ in range(40): capi.disk_image_device(d,bf.ptr,bg.ptr)
    e1.record(); print("direct" if direct else "default", "%.4f ms" % (e0.elapsed_ms(e1)/40))
