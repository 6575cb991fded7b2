import sys, time; sys.path.insert(0,'.')
import numpy as np, lws_amd
from bench import synth_magnitudes
B=256
M = synth_magnitudes(B, 500, 513, 20260928).astype(np.complex128)
thr10 = lws_amd.get_thresholds(10,1,0.1,1)
p = lws_amd.lws(1024, 256, mode='music'); plan = p.plan()
for i in range(2):
    out = plan.online(M, thr10, 3, 4.0)
print(plan.last_kernel())
