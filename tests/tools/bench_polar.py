import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
n=2048; d=capi.image_desc(n,n,0.9,70/180*math.pi,pol_degree=0.1)
st=capi.DeviceBuffer(3*n*n*8); ch=capi.DeviceBuffer(n*n*8)
out=[]
for want_chi in (0,1):
    fn=lambda: capi.disk_image_polarized_device(d, st.ptr, ch.ptr if want_chi else None)
    import time; t0=time.time()
    while time.time()-t0 < 0.5: fn()        # working clock (an idle GPU runs its first ~50 ms slower)
    capi.synchronize(); e0=capi.Event(); e1=capi.Event(); e0.record()
    for _ in range(100): fn()
    e1.record(); ms=e0.elapsed_ms(e1)/100
    S=st.to_numpy(np.float64,(3,n,n))
    out.append("%s %.4f ms sumQ %.10e sumU %.10e"%("chi" if want_chi else "nochi",ms,S[1].sum(),S[2].sum()))
print(" | ".join(out))
