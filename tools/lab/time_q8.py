import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch, lws_amd
from lws_amd import _capi
B, T, iters, F = 256, 500, 100, 513
p = lws_amd.lws(1024, 128)
rng = np.random.default_rng(0)
S = torch.from_numpy((rng.standard_normal((B,T,F)) + 1j*rng.standard_normal((B,T,F))).astype(np.complex64)).cuda()
thr = np.zeros(iters)
plan = _capi.Plan(F, p.W)
st = torch.cuda.current_stream().cuda_stream
plan.batch_dev(S.data_ptr(), B, T, thr, stream=st); torch.cuda.synchronize()
for _ in range(2):
    t0=time.perf_counter(); plan.batch_dev(S.data_ptr(), B, T, thr, stream=st); torch.cuda.synchronize(); dt=time.perf_counter()-t0
print("%s %.2f ms kernel %.2f ms" % (plan.last_kernel()["name"], dt*1e3, plan.last_kernel()["ms"]), flush=True)
