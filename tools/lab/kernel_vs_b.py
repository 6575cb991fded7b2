import time, numpy as np, torch, lws_amd
T, F = 500, 513
p = lws_amd.lws(1024, 256); plan = p.plan()
thr = np.zeros(100)
for B in (8, 16, 32, 64, 96, 128, 256):
    d = torch.rand((B, T, F), device="cuda").to(torch.complex64)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan.batch_dev(d.data_ptr(), B, T, thr); torch.cuda.synchronize()
        w = 1e3 * (time.perf_counter() - t0)
    print(B, "wall %.2f ms" % w, plan.last_kernel(), "-> per 256: %.1f ms" % (w * 256 / B))
