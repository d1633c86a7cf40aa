import numpy as np,sys
a=np.load(sys.argv[1]); b=np.load(sys.argv[2])
for k,nm in enumerate("IQUc"):
    x,y=a[k],b[k]
    bad=~((x==y)|(np.isnan(x)&np.isnan(y)))
    rel=np.abs(x-y)/np.maximum(np.abs(y),1e-300)
    big=bad&(rel>1e-6)
    print(nm,"differs",bad.sum(),"rel>1e-6",big.sum(), "sum",np.nansum(x),np.nansum(y))
    if big.sum():
        idx=np.argwhere(big)[:5]; print(idx.tolist(), [ (x[i,j],y[i,j]) for i,j in idx])
