"""config 3 (run_lws(mode='music'): no-future -> online -> batch) through the host-array entry point, numpy in / numpy out"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import lws_amd
B, T, F = 256, 500, 513
p = lws_amd.lws(1024, 256, mode="music")
rng = np.random.default_rng(1)
M = np.abs(rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))).astype(np.complex128)
for env in ({}, {"LWS_HOST_CHUNK_BINS": str(1 << 30)}, {"LWS_HOST_MONOLITHIC": "1"}):
    for k in ("LWS_HOST_CHUNK_BINS", "LWS_HOST_MONOLITHIC"): os.environ.pop(k, None)
    os.environ.update(env)
    outs = []
    for i in range(3):
        t0 = time.perf_counter(); o = p.run_lws(M); dt = 1e3 * (time.perf_counter() - t0)
        outs.append(dt); keep = o
    print(env, " ".join("%.1f" % x for x in outs), "ms", p.plan().last_kernel()["name"], flush=True)
