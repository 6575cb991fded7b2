import sys, time, os; sys.path.insert(0,'.')
import numpy as np, lws_amd
def run(fs,sh,B,T,it,nwg,seed=0):
    os.environ["LWS_SYSTOLIC_NWG"]=str(nwg)
    F=fs//2+1
    rng=np.random.default_rng(seed)
    S=np.abs(rng.standard_normal((B,T,F))+1j*rng.standard_normal((B,T,F))).astype(np.complex128)
    p = lws_amd.lws(fs, sh); plan=p.plan()
    out=plan.batch(S,np.zeros(it)); k=plan.last_kernel()
    return out,k
for (fs,sh,B,T,it,nwg) in [(1024,256,16,300,60,4),(1024,256,32,300,60,4),(1024,256,40,300,60,4),(1024,256,48,300,60,4),(1024,256,56,300,60,4),(1024,256,60,300,60,4),(1024,256,64,300,60,4),(1024,256,64,300,60,2),(1024,256,128,300,60,2),(1024,256,100,300,60,2)]:
    r1,k1=run(fs,sh,B,T,it,1); rn,kn=run(fs,sh,B,T,it,nwg)
    d=np.abs(r1-rn)
    bad=[b for b in range(B) if not np.array_equal(r1[b],rn[b])]
    print(fs,sh,"B",B,"T",T,"it",it,"nwg",nwg,"ms %.2f -> %.2f"%(k1['ms'],kn['ms']),"identical",np.array_equal(r1,rn),"max diff %.2e"%d.max(),"bad spectrograms",bad[:10],len(bad),flush=True)
