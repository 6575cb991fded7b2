"""Times the fp64 batch engines on device-resident spectrograms (kernel time between the plan's events, and wall time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lws_amd
from lws_amd import _capi


def t(fsize, fshift, B, T, iters, generic=False):
    F = fsize // 2 + 1
    p = lws_amd.lws(fsize, fshift)
    S = torch.from_numpy(np.abs(np.random.default_rng(0).standard_normal((B, T, F))).astype(np.complex128)).cuda()
    plan = _capi.Plan(F, p.W, precision="fp64", force_generic=generic)
    thr = np.zeros(iters)
    plan.batch_dev(S.data_ptr(), B, T, thr); torch.cuda.synchronize()
    ms, wall = [], []
    for _ in range(2):
        t0 = time.perf_counter()
        plan.batch_dev(S.data_ptr(), B, T, thr); torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e3); ms.append(plan.last_kernel()["ms"])
    k = plan.last_kernel()
    n = B * T * F * iters
    print("%5d/%4d B=%d T=%d F=%4d iters=%d %-22s kernels %s ms, wall %s ms, launches %d, ps/bin-sweep %.1f"
          % (fsize, fshift, B, T, F, iters, k["name"], " ".join("%.1f" % m for m in ms), " ".join("%.1f" % m for m in wall),
             k.get("launches", -1), min(ms) * 1e9 / n), flush=True)
    plan.close()


if __name__ == "__main__":
    if "--wide-one" in sys.argv:   # profiling target of tools/profile_sys64.sh <tag> --wide-one
        t(2048, 512, 256, 250, 40)
        sys.exit(0)
    if "--wide" in sys.argv:   # 128 frames in flight (round 5): 2048-point frames, and the shapes the 64-lane geometry held one slot of
        t(2048, 512, 256, 250, 40)
        t(2048, 512, 256, 250, 40, generic=True)
        t(2048, 1024, 256, 250, 40)
        t(1536, 384, 256, 250, 40)
        t(1200, 300, 256, 500, 40)
        t(1100, 275, 256, 500, 40)
        t(4096, 1024, 256, 250, 20)      # 256 frames in flight
        t(4096, 1024, 256, 250, 20, generic=True)
        t(4096, 2048, 256, 250, 20)
        sys.exit(0)
    t(1024, 256, 256, 500, 100)
    if "--one" in sys.argv:
        sys.exit(0)
    if "--generic" in sys.argv:
        t(1024, 256, 256, 500, 100, generic=True)
    t(512, 128, 256, 500, 100)
    t(512, 128, 512, 500, 100)       # two 257-bin spectrograms per workgroup
    t(400, 100, 512, 500, 100)
    t(1024, 512, 256, 500, 100)
