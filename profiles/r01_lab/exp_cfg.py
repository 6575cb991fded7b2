import sys, time; sys.path.insert(0,'.')
import numpy as np, lws_amd
from bench import synth_magnitudes
B=int(sys.argv[1]) if len(sys.argv)>1 else 256
p = lws_amd.lws(1024, 256, mode='music')
M = synth_magnitudes(B, 500, 513, 20260928).astype(np.complex128)
plan = p.plan()
thr1 = lws_amd.get_thresholds(1,1,0.1,1); thr10 = lws_amd.get_thresholds(10,1,0.1,1); thr100 = lws_amd.get_thresholds(100,100,0.1,1)
for name, fn in [("nofuture(1 it, Q4 compat)", lambda: plan.nofuture(M, thr1, wsel=1)),
                 ("online(10 it, LA=3)", lambda: plan.online(M, thr10, 3, 4.0)),
                 ("batch(100 it default)", lambda: plan.batch(M, thr100)),
                 ("run_lws music", lambda: p.run_lws(M))]:
    t=time.time(); out = fn(); dt=time.time()-t
    k = plan.last_kernel()
    print(f"{name}: B={B} wall {dt:.3f}s  last kernel {k}", flush=True)
