#!/usr/bin/env python3
"""Premise of the concurrent-chunk host pipeline: config 2 as four launches of 64 spectrograms, one workgroup
per spectrogram, on four streams with a plan each -- do they share the chip (total ~ one launch of 256) or queue up?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import lws_amd
from lws_amd import _capi

B, T, F, K = 256, 500, 513, int(sys.argv[1]) if len(sys.argv) > 1 else 4
p = lws_amd.lws(1024, 256)
thr = np.zeros(100)
x = torch.rand((B, T, F), device="cuda").to(torch.complex64)
whole = _capi.Plan(F, p.W)
def run_whole():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    whole.batch_dev(x.data_ptr(), B, T, thr, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0)
for _ in range(3): w = run_whole()
print("one launch of 256: %.2f ms (%s)" % (w, whole.last_kernel()["name"]))
n = B // K
def seq():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        whole.batch_dev(x[k * n:(k + 1) * n].data_ptr(), n, T, thr, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0)
for _ in range(3): s = seq()
print("%d launches of %d in sequence (several workgroups each): %.2f ms" % (K, n, s))
os.environ["LWS_SYSTOLIC_NWG"] = "1"
plans = [_capi.Plan(F, p.W) for _ in range(K)]
streams = [torch.cuda.Stream() for _ in range(K)]
def conc(stagger_ms=0.0):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        plans[k].batch_dev(x[k * n:(k + 1) * n].data_ptr(), n, T, thr, stream=streams[k].cuda_stream)
        if stagger_ms: time.sleep(stagger_ms * 1e-3)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0)
for _ in range(3): c = conc()
print("%d launches of %d on %d streams, one workgroup per spectrogram: %.2f ms" % (K, n, K, c))
for _ in range(2): c = conc(1.5)
print("... started 1.5 ms apart: %.2f ms (kernel of the last: %.2f ms)" % (c, plans[-1].last_kernel()["ms"]))
one = plans[0]
torch.cuda.synchronize(); t0 = time.perf_counter()
one.batch_dev(x[:n].data_ptr(), n, T, thr, stream=streams[0].cuda_stream)
torch.cuda.synchronize()
print("one launch of %d alone, one workgroup per spectrogram: %.2f ms" % (n, 1e3 * (time.perf_counter() - t0)))
