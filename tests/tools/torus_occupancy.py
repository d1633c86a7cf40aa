import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
import test_gpu_raytrace as T
n=1024; N=n*n
dd=T.torus_desc(capi,n,0.9,70.0,r0=100.0,precision=float(sys.argv[1]) if len(sys.argv)>1 else 1.0,max_steps=100000)
sb=capi.DeviceBuffer(N*40); steps=capi.DeviceBuffer(N*4); dbg=capi.DeviceBuffer(N*32)
dbg.from_numpy(np.zeros(N*4,dtype=np.float64))
capi.torus_image_device(dd, sb.ptr, aux={"steps":steps.ptr,"k_end":dbg.ptr}); capi.synchronize()
c=dbg.to_numpy(np.uint64,(N*4,))[:6]
print("V batches %d avg lanes %.1f | R batches %d avg lanes %.1f"%(c[0], c[1]/max(c[0],1), c[2], c[3]/max(c[2],1)))
t=dbg.to_numpy(np.uint64,(N*4,))[16:16+3*3072].reshape(-1,3).astype(np.float64)
t=t[t[:,0]>0]; t0=t[:,0].min()
dr=t[:,1][t[:,1]>0]
print("waves %d | cursor exhausted (first wave) at %.2f ms, (median wave) %.2f ms | exits: first %.2f median %.2f p90 %.2f last %.2f ms"%(
    len(t),(dr.min()-t0)/1e5,(np.median(dr)-t0)/1e5,(t[:,2].min()-t0)/1e5,(np.median(t[:,2])-t0)/1e5,(np.percentile(t[:,2],90)-t0)/1e5,(t[:,2].max()-t0)/1e5))
s=steps.to_numpy(np.int32,(N,))
print("steps: mean %.1f p50 %d p99 %d p99.9 %d max %d"%(s.mean(),np.percentile(s,50),np.percentile(s,99),np.percentile(s,99.9),s.max()))
