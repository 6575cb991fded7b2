"""Where does the online LDS engine differ from the fp64 oracle?  PYTHONPATH=. python tools/lab/online_check.py fsize fshift T LA iters [layout]"""
import os, sys
import numpy as np
sys.path.insert(0, "tests")
import lws_amd
from lws_amd import _capi
from oracle.oracle import Oracle
fsize, fshift, T, LA, iters = [int(x) for x in sys.argv[1:6]]
if len(sys.argv) > 6: os.environ["LWS_ONLINE_LAYOUT"] = sys.argv[6]
rng = np.random.default_rng(fsize + T)
p = lws_amd.lws(fsize, fshift, mode="music")
F = fsize // 2 + 1
S = rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F))
thr = [0.0] * iters
W = (p.W, p.W_ai, p.W_af)
plan = _capi.Plan(F, *W)
np.set_printoptions(linewidth=250, formatter={"float": lambda x: "%.1e" % x})
for it in range(1, iters + 1):
    r2 = Oracle().online_lws(S, *W, thr[:it], LA, fshift)
    o2 = plan.online(S, thr[:it], LA, fsize / fshift)
    e2 = np.abs(o2 - r2)
    print("iters=%d %s: median %.2e max %.2e" % (it, plan.last_kernel()["name"], np.median(e2), e2.max()))
    print("  per frame max:", e2.max(axis=1))
    bad = np.argwhere(e2 > 1e-5)
    print("  bins with err > 1e-5: %d; first 40 (frame, bin):" % len(bad), [tuple(int(v) for v in x) for x in bad[:40]])
if os.environ.get("DUMP"):
    r2 = Oracle().online_lws(S, *W, thr[:1], LA, fshift)
    o2 = plan.online(S, thr[:1], LA, fsize / fshift)
    np.set_printoptions(linewidth=250, precision=4, suppress=True)
    print("ref  frame0 tail:", r2[0, -10:])
    print("out  frame0 tail:", o2[0, -10:])
    print("in   frame0 tail:", S[0, -10:])
    print("|out| - |in| tail:", np.abs(o2[0, -10:]) - np.abs(S[0, -10:]))

if os.environ.get("LWS_HIP_LIB") and os.environ.get("LABDBG"):
    import ctypes as C
    lib = C.CDLL(os.environ["LWS_HIP_LIB"])
    buf = (C.c_ulonglong * 256)()
    lib.lws_lab_read(buf, 256)
    n = buf[0]
    print("mismatches between sums read before / after the barrier (accumulated over the calls above):", n)
    print([(buf[i] >> 32, (buf[i] >> 8) & 0xffffff, buf[i] & 0xff) for i in range(1, min(int(n), 60) + 1)])
