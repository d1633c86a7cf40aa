import sys, math, hashlib, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
n=4096
d=capi.image_desc(n,n,0.998,70/180*math.pi)
bf=capi.DeviceBuffer(n*n*4); bg=capi.DeviceBuffer(n*n*4)
for _ in range(400 if '--warm' in sys.argv else 5): capi.disk_image_device(d,bf.ptr,bg.ptr)   # --warm: ~0.17 s first, working clock
capi.synchronize(); e0=capi.Event(); e1=capi.Event(); e0.record()
for _ in range(40): capi.disk_image_device(d,bf.ptr,bg.ptr)
e1.record(); ms=e0.elapsed_ms(e1)/40
g=bg.to_numpy(np.float32,(n,n)); f=bf.to_numpy(np.float32,(n,n))
print("%.4f ms  %.3e rays/s  hits %d  md5 %s" % (ms, n*n/ms*1e3, int((g>0).sum()), hashlib.md5(g.tobytes()+f.tobytes()).hexdigest()[:12]))
