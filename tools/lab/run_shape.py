"""one batch call of a given shape (for profiler passes): python tools/lab/run_shape.py fsize fshift B T iters"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch, lws_amd
from lws_amd import _capi
fsize, fshift, B, T, iters = [int(v) for v in sys.argv[1:6]]
F = fsize // 2 + 1
p = lws_amd.lws(fsize, fshift)
S = torch.from_numpy(np.abs(np.random.default_rng(0).standard_normal((B, T, F))).astype(np.complex64)).cuda()
plan = _capi.Plan(F, p.W)
for _ in range(2):
    plan.batch_dev(S.data_ptr(), B, T, np.zeros(iters)); torch.cuda.synchronize()
print(plan.last_kernel())
