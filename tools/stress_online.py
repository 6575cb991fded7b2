"""Random shapes / look-aheads / iteration counts: the online LDS engine's serial-taps variant (either layout) must reproduce the
generic engine's fp32 result bit for bit (schedule, frame ring, sweep slots), the production variant the same magnitudes.
usage: PYTHONPATH=. python tools/stress_online.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import lws_amd
from lws_amd import _capi

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cfgs = [(64, 16), (64, 32), (64, 8), (128, 32), (256, 64), (256, 32), (512, 128), (512, 64), (1024, 256), (1024, 512), (1024, 128),
        (1000, 250), (2048, 512), (60, 15), (4096, 1024), (3000, 750), (4096, 2048)]
bad = 0
for it in range(cases):
    fs, sh = cfgs[rng.integers(len(cfgs))]
    F = fs // 2 + 1
    T = int(rng.integers(1, 60 if fs <= 2048 else 24))
    LA = int(rng.integers(0, 8))
    iters = int(rng.integers(1, 9))
    B = int(rng.integers(1, 3))
    layout = str(rng.choice(["2", "3"]))
    L = int(rng.choice([5, 5, 5, 1, 2, 3, 4]))          # (stencils narrower than the kernel's run on the fourth layout with zero weights)
    layout = "4" if L != 5 else str(rng.choice(["2", "3", "4"]))
    p = lws_amd.lws(fs, sh, L=L, mode="music")
    W = (p.W, p.W_ai, p.W_af)
    S = rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))
    if rng.random() < 0.3:
        S = np.abs(S) + 0j
    thr = lws_amd.get_thresholds(iters, 1.0, 0.1, 1)
    def run(**kw):
        plan = _capi.Plan(F, *W, **kw)
        out = plan.online(S, thr, LA, fs / sh)
        name = plan.last_kernel()["name"]
        plan.close()
        return out, name
    ref, _ = run(force_generic=True)
    os.environ["LWS_ONLINE_LAYOUT"] = layout
    os.environ["LWS_ONLINE_SERIAL_TAPS"] = "1"
    ser, name = run()
    del os.environ["LWS_ONLINE_SERIAL_TAPS"]
    prod, _ = run()
    del os.environ["LWS_ONLINE_LAYOUT"]
    same = np.array_equal(ser, ref) if name.startswith("online_lds") else True
    mag = np.abs(np.abs(prod) - np.abs(ref)).max() / np.abs(S).max()
    ok = same and mag < 2e-6 and np.isfinite(prod).all()
    bad += not ok
    print(f"{fs:5d} {sh:4d} B={B} T={T:3d} LA={LA} it={iters} layout={layout} {name:18s} serial==generic {same} mag {mag:.1e}{'' if ok else '   <<<<<<<<'}", flush=True)
print("failures", bad)
sys.exit(1 if bad else 0)
