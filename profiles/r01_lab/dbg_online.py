import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, lws_amd
from lws_amd import _capi
from oracle.oracle import Oracle
from conftest import load_golden
oracle=Oracle()
h, g = load_golden("helpers.npz"), load_golden("wrappers.npz")
for tag in ["64_16","64_32","64_8"]:
  for T in (1,2,3,7,24):
    for LA in (0,1,3,5):
        W = (h[f"W_{tag}"], h[f"W_ai_{tag}"], h[f"W_af_{tag}"])
        fsize, fshift = [int(v) for v in tag.split("_")]
        S = g[f"S_{tag}"][:T]; F=S.shape[1]; thr=[0.6,0.2,0.0]
        ref = oracle.online_lws(S, *W, thr, LA, fshift)
        plan=_capi.Plan(F,*W); out=plan.online(S,thr,LA,fsize/fshift); plan.close()
        plan=_capi.Plan(F,*W,force_generic=True); outg=plan.online(S,thr,LA,fsize/fshift); plan.close()
        err=np.abs(out-ref); eg=np.abs(outg-ref); sc=np.mean(np.abs(S))
        print(tag,T,LA,"lds: med %.1e max %.1e rel %.1e | gen: med %.1e max %.1e rel %.1e"%(np.median(err)/sc,err.max()/sc,np.linalg.norm(err)/np.linalg.norm(ref),np.median(eg)/sc,eg.max()/sc,np.linalg.norm(eg)/np.linalg.norm(ref)))
