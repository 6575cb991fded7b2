import sys, time; sys.path.insert(0,'.')
import numpy as np, lws_amd
from bench import synth_magnitudes
B=int(sys.argv[1]) if len(sys.argv)>1 else 256
M = synth_magnitudes(B, 500, 513, 20260928).astype(np.complex128)
thr10 = lws_amd.get_thresholds(10,1,0.1,1)
outs={}
for fg in (False, True):
    p = lws_amd.lws(1024, 256, mode='music', force_generic=fg)
    plan = p.plan()
    t=time.time(); out = plan.online(M, thr10, 3, 4.0); dt=time.time()-t
    outs[fg]=out
    print(f"force_generic={fg} online(10 it, LA=3): B={B} wall {dt:.3f}s  last kernel {plan.last_kernel()}", flush=True)
d=np.abs(outs[True]-outs[False]); print("lds vs generic: max", d.max(), "rel-l2", np.linalg.norm(d)/np.linalg.norm(outs[True]), "median", np.median(d))
p = lws_amd.lws(1024, 256, mode='music')
t=time.time(); out=p.run_lws(M); print("run_lws music wall", time.time()-t)
