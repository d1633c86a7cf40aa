import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
n=4096; y0=int(sys.argv[1]); y1=int(sys.argv[2])
d=capi.image_desc(n,n,0.998,70/180*math.pi,y0=y0,y1=y1)
rows=y1-y0
bf=capi.DeviceBuffer(rows*n*4); bg=capi.DeviceBuffer(rows*n*4)
for _ in range(3): capi.disk_image_device(d,bf.ptr,bg.ptr)
capi.synchronize(); e0=capi.Event(); e1=capi.Event(); e0.record()
for _ in range(20): capi.disk_image_device(d,bf.ptr,bg.ptr)
e1.record(); ms=e0.elapsed_ms(e1)/20
g=bg.to_numpy(np.float32,(rows,n))
print("rows [%d,%d): %.4f ms  hits %d"%(y0,y1,ms,int((g>0).sum())))
