import sys, time; sys.path.insert(0,'.')
import numpy as np, lws_amd, torch
from bench import synth_magnitudes
B=int(sys.argv[1]) if len(sys.argv)>1 else 256
M = synth_magnitudes(B, 500, 513, 20260928).astype(np.complex128)
def ktime(plan): return plan.last_kernel()
for prec in ('fp32','fp64'):
    p = lws_amd.lws(1024, 256, mode='music', precision=prec, force_generic=True)
    plan = p.plan()
    thr1 = lws_amd.get_thresholds(1,1,0.1,1); thr10 = lws_amd.get_thresholds(10,1,0.1,1)
    z10 = np.zeros(10)
    for name, fn in [("batch dense 10 it", lambda: plan.batch(M, z10)),
                     ("nofuture(1 it, Q4 compat)", lambda: plan.nofuture(M, thr1, wsel=1)),
                     ("online(10 it, LA=3)", lambda: plan.online(M, thr10, 3, 4.0))]:
        t=time.time(); out = fn(); dt=time.time()-t
        print(f"{prec} {name}: B={B} wall {dt:.3f}s  last kernel {ktime(plan)}", flush=True)
