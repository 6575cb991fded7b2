# per-wave wait / work clocks of workgroup 0 (variant built with -DLWS_DBG_TIMING), one dense 100-sweep launch
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import lws_amd
from lws_amd import _capi
p = lws_amd.lws(1024, 256)
B, T, F = 256, 500, 513
plan = _capi.Plan(F, p.W, p.W_ai, p.W_af)
S = torch.rand(B, T, F, device="cuda").to(torch.complex64)
thr = np.zeros(100)
plan.batch_dev(S.data_ptr(), B, T, thr)
torch.cuda.synchronize()
print(plan.last_kernel())
