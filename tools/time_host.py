"""Wall time of the host-array entry point plan.batch(numpy complex128) on BASELINE config 2's shape (what lws.lws().batch_lws /
run_lws cost a caller who holds numpy arrays), against the device-resident call.  PYTHONPATH=. python tools/time_host.py [B] [iters]"""
import sys, time
import numpy as np, torch
import lws_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
T, F = 500, 513
p = lws_amd.lws(1024, 256)
plan = p.plan()
rng = np.random.default_rng(1)
M = np.abs(rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))).astype(np.complex128)
thr = np.zeros(iters)
outs = []
for rep in range(4):
    t0 = time.perf_counter()
    outs.append(plan.batch(M, thr))      # (kept: freeing a 1 GB result -- 256K pages back to the OS -- costs 20-40 ms by itself)
    t1 = time.perf_counter()
    out = outs[-1]
    if len(outs) > 2: outs.pop(0)
    print("host arrays: plan.batch(%dx%dx%d complex128, %d sweeps) wall %.1f ms  kernel of the last chunk %s" % (B, T, F, iters, 1e3 * (t1 - t0), plan.last_kernel()), flush=True)
d = torch.from_numpy(M.astype(np.complex64)).cuda()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan.batch_dev(d.data_ptr(), B, T, thr); torch.cuda.synchronize()
    print("device resident: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
ref = d.cpu().numpy()
print("max |host path - device path| = %.3e; magnitudes kept: %.2e" % (np.abs(out - ref).max(), np.abs(np.abs(out) - np.abs(M)).max()))
# pinned copy rate of this box
h = torch.empty(1 << 28, dtype=torch.uint8).pin_memory(); g = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.copy_(h, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
    h.copy_(g, non_blocking=True); torch.cuda.synchronize(); t2 = time.perf_counter()
print("pinned copy rate: H2D %.1f GB/s, D2H %.1f GB/s" % (0.268 / (t1 - t0), 0.268 / (t2 - t1)))
