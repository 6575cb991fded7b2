"""Time one batch call of config 2's volume at hop 256 (Q = 4) and hop 128 (Q = 8), three times each: for A/B runs of kernel
variants on ONE box (LWS_HIP_LIB=... selects the library; boxes of the pool differ by ~2 %).  usage: PYTHONPATH=. python tools/time_batch_shapes.py"""
import numpy as np, time, torch
import lws_amd
from lws_amd import _capi
B,T,F=256,500,513
for hop in (256,128):
    p=lws_amd.lws(1024,hop)
    pl=p.plan()
    g=torch.Generator(device='cuda'); g.manual_seed(1)
    mag=torch.rand((B,T,F),device='cuda',generator=g)
    S=torch.complex(mag,torch.zeros_like(mag)).contiguous()
    thr=np.zeros(100)
    out=torch.empty_like(S)
    for it in range(3):
        torch.cuda.synchronize(); t0=time.time()
        S.copy_(torch.complex(mag,torch.zeros_like(mag))); torch.cuda.synchronize(); t0=time.time(); pl.batch_dev(S.data_ptr(), B, T, thr)
        torch.cuda.synchronize(); t1=time.time()
        print(hop, pl.last_kernel(), (t1-t0)*1e3)
