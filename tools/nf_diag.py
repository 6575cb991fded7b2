import numpy as np, lws_amd
from lws_amd import _capi
from oracle.oracle import Oracle
orc = Oracle()
def rel(a,b): return np.linalg.norm(a-b)/np.linalg.norm(b)
for fsize,fshift,T,n_it in ((400,160,40,2),(400,160,40,1),(512,128,40,2),(48,16,40,2)):
    p = lws_amd.lws(fsize, fshift, nofuture_q4_compat=False)
    F = fsize//2+1
    rng = np.random.default_rng(fsize+T)
    S = rng.standard_normal((1,T,F)) + 1j*rng.standard_normal((1,T,F))
    thr = np.linspace(0.5,0.0,n_it)
    plan = _capi.Plan(F,p.W,p.W_ai,p.W_af, nofuture_q4_compat=False); gen = _capi.Plan(F,p.W,p.W_ai,p.W_af,force_generic=True, nofuture_q4_compat=False)
    p64 = _capi.Plan(F,p.W,p.W_ai,p.W_af,precision="fp64", nofuture_q4_compat=False)
    out = plan.nofuture(S,thr,wsel=1)[0]; name=plan.last_kernel()["name"]
    outg = gen.nofuture(S,thr,wsel=1)[0]; o64 = p64.nofuture(S,thr,wsel=1)[0]
    ref = orc.nofuture_lws(S[0],p.W_ai,thr,compat=False)
    refw = orc.nofuture_lws(S[0],p.W_ai.astype(np.complex64).astype(np.complex128),thr,compat=False)
    print(fsize,fshift,n_it,name,"lds",rel(out,ref),"generic",rel(outg,ref),"fp64",rel(o64,ref),"oracle w/ fp32 weights",rel(refw,ref), "lds vs generic", rel(out,outg))
    print("   per-frame lds:", ["%.1e"%rel(out[f],ref[f]) for f in (0,3,5,10,20,30,39)], " generic:", ["%.1e"%rel(outg[f],ref[f]) for f in (0,3,5,10,20,30,39)])
