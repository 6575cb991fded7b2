import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, lws_amd
from lws_amd import _capi
def chk(fsize, fshift, T, n_it, B=1):
    F = fsize//2+1
    p = lws_amd.lws(fsize, fshift)
    rng = np.random.default_rng(T+n_it)
    S = rng.standard_normal((B,T,F)) + 1j*rng.standard_normal((B,T,F))
    thr = np.zeros(n_it)
    a = _capi.Plan(F, p.W); out = a.batch(S, thr); name = a.last_kernel()["name"]
    os.environ["LWS_SYSTOLIC_NO_SHORT"]="1"
    g = _capi.Plan(F, p.W); ref = g.batch(S, thr); name2 = g.last_kernel()["name"]
    del os.environ["LWS_SYSTOLIC_NO_SHORT"]
    d = np.abs(out-ref)
    bad = np.argwhere(d > 1e-3*np.abs(S).mean())
    print(fsize,fshift,"T",T,"it",n_it,name,"vs",name2,"rel",np.linalg.norm(out-ref)/np.linalg.norm(ref),"nbad",len(bad), "first bad", bad[:4].tolist(), flush=True)
for args in ((512,128,1,1),(512,128,5,1),(512,128,40,1),(512,128,40,3),(512,128,40,14),(512,128,40,15),(512,128,70,30),
             (64,16,1,1),(64,16,5,1),(64,16,20,1),(64,16,20,3),(64,16,40,24),(64,16,40,25),(256,64,40,30)):
    chk(*args)
for args in ((512,128,3,40),(512,128,20,29),(512,128,26,100),(64,16,3,100),(64,16,90,100),(64,16,100,49),(256,64,200,100),(128,32,33,77),(512,256,9,33),(256,128,50,60),(252,126,40,50),(500,125,40,31),(120,30,70,50),(200,50,140,60)):
    chk(*args, B=2)
