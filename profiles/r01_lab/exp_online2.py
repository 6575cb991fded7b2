import sys, time; sys.path.insert(0,'.')
import numpy as np, lws_amd
from lws_amd import _capi
rng=np.random.default_rng(5)
for (fs,sh,T,LA,it) in [(64,16,24,3,3),(128,32,40,3,4),(256,64,60,3,5),(512,128,100,3,10),(1024,256,120,3,10),(1024,256,120,5,4),(1024,512,80,3,10),(1024,128,60,2,6)]:
    p=lws_amd.lws(fs,sh,mode='music'); F=fs//2+1
    S=rng.standard_normal((2,T,F))+1j*rng.standard_normal((2,T,F))
    thr=lws_amd.get_thresholds(it,1,0.1,1)
    res={}
    for fg in (False,True):
        plan=_capi.Plan(F,p.W,p.W_ai,p.W_af,force_generic=fg)
        res[fg]=plan.online(S,thr,LA,fs/sh); nm=plan.last_kernel()['name'] if not fg else nm; plan.close()
    plan=_capi.Plan(F,p.W,p.W_ai,p.W_af,precision='fp64'); r64=plan.online(S,thr,LA,fs/sh); plan.close()
    d=np.abs(res[True]-res[False]); e1=np.abs(res[False]-r64); e2=np.abs(res[True]-r64)
    print(fs,sh,T,LA,it,nm,"lds-vs-gen max %.2e"%d.max(),"lds-vs-fp64 max %.2e relL2 %.2e"%(e1.max(),np.linalg.norm(e1)/np.linalg.norm(r64)),"gen-vs-fp64 max %.2e relL2 %.2e"%(e2.max(),np.linalg.norm(e2)/np.linalg.norm(r64)),flush=True)
