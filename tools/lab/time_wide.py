import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch, lws_amd
from lws_amd import _capi
B, T, iters = 64, 6000, 60
F = 1025
p = lws_amd.lws(2048, 512)
rng = np.random.default_rng(0)
S = torch.from_numpy((rng.standard_normal((B,T,F)) + 1j*rng.standard_normal((B,T,F))).astype(np.complex64)).cuda()
thr = np.zeros(iters)
plan = _capi.Plan(F, p.W)
st = torch.cuda.current_stream().cuda_stream
plan.batch_dev(S.data_ptr(), B, T, thr, stream=st); torch.cuda.synchronize()
for _ in range(3):
    t0=time.perf_counter(); plan.batch_dev(S.data_ptr(), B, T, thr, stream=st); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print("%s %.2f ms  %.3f ps/bin-sweep  kernel %.2f ms  %.1f %% of 8 TB/s" % (plan.last_kernel()["name"], dt*1e3, dt/(B*T*F*iters)*1e12, plan.last_kernel()["ms"], 20.0*B*T*F*iters/(plan.last_kernel()["ms"]*1e-3)/8e12*100), flush=True)
