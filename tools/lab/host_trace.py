import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import lws_amd
from lws_amd import _capi
B, T, F, iters = 256, 500, 513, 100
p = lws_amd.lws(1024, 256); plan = p.plan(); lib = _capi.load()
rng = np.random.default_rng(1)
M = np.abs(rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))).astype(np.complex128)
thr = np.zeros(iters)
out = np.empty_like(M); out[:] = 0
def call():
    t0 = time.perf_counter()
    _capi.check(lib.lws_batch_lws(plan._h, 0, M.ctypes.data, out.ctypes.data, B, T, thr.ctypes.data, iters))
    return 1e3 * (time.perf_counter() - t0)
call(); call()
os.environ["LWS_HOST_TRACE"] = "1"
print("wall %.1f ms" % call(), flush=True)
