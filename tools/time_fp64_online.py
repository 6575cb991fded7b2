"""The online stage of an fp64 plan on config 3's shape (profiling target: tools/pmc_sq_kernel.sh r05_online64 k_online64 1 python tools/time_fp64_online.py);
with arguments `fsize hop T [generic]`: another frame size (2048-point frames keep their magnitudes in memory, lws_online64.hip)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch, lws_amd
fs, hop, T = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 256, 500)
B, F = 256, fs // 2 + 1
pm = lws_amd.lws(fs, hop, mode="music", precision="fp64", force_generic="generic" in sys.argv)
thr = lws_amd.get_thresholds(pm.online_iterations, pm.online_alpha, pm.online_beta, pm.online_gamma)
M = np.abs(np.random.default_rng(0).standard_normal((B, T, F)) + 1j * np.random.default_rng(1).standard_normal((B, T, F))).astype(np.complex128)
d = torch.from_numpy(M).cuda()
for rep in range(2):
    d.copy_(torch.from_numpy(M)); torch.cuda.synchronize(); t0 = time.perf_counter()
    pm.plan().online_dev(d.data_ptr(), B, T, thr, pm.look_ahead, fs / hop); torch.cuda.synchronize()
    print("lws(%d,%d) %d x %d x %d online fp64: %.1f ms (%s)" % (fs, hop, B, T, F, 1e3 * (time.perf_counter() - t0), pm.plan().last_kernel()["name"]))
