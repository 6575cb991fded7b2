import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import lws_amd
from lws_amd import _capi
B, T, F, iters = 256, 500, 513, 100
p = lws_amd.lws(1024, 256); plan = p.plan(); lib = _capi.load()
rng = np.random.default_rng(1)
M = np.abs(rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))).astype(np.complex128)
thr = np.zeros(iters)
out = np.empty_like(M); out[:] = 0
def call(reuse):
    o = out if reuse else np.empty_like(M)
    t0 = time.perf_counter()
    _capi.check(lib.lws_batch_lws(plan._h, 0, M.ctypes.data, o.ctypes.data, B, T, thr.ctypes.data, iters))
    return 1e3 * (time.perf_counter() - t0)
for env in ({}, {"LWS_HOST_THREADS": "8"}, {"LWS_HOST_THREADS": "16"}, {"LWS_HOST_THREADS": "64"}, {"LWS_HOST_CHUNK_BINS": str(16 << 20)}, {"LWS_HOST_CHUNK_BINS": str(32 << 20)},
            {"LWS_HOST_CHUNK_BINS": str(16 << 20), "LWS_HOST_THREADS": "64"}):
    for k in ("LWS_HOST_THREADS", "LWS_HOST_CHUNK_BINS"): os.environ.pop(k, None)
    os.environ.update(env)
    call(True)
    print(env, "reused output: %.1f %.1f ms | fresh output: %.1f %.1f ms" % (call(True), call(True), call(False), call(False)), flush=True)
# bit-identity of the kernel across workgroups-per-spectrogram (a 32-spectrogram call against the first 32 of a 256 call)
d = torch.from_numpy(M.astype(np.complex64)).cuda(); d2 = d[:32].clone()
th = lws_amd.get_thresholds(100, 100, 0.1, 1)
plan.batch_dev(d.data_ptr(), B, T, th); plan.batch_dev(d2.data_ptr(), 32, T, th); torch.cuda.synchronize()
print("32-spectrogram launch == first 32 of the 256 launch (default schedule):", bool(torch.equal(d[:32], d2)), plan.last_kernel())
