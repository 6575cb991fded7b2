import numpy as np, sys
import lws_amd
from lws_amd import _capi
fsize, hop, T, n_it = 64, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1
p = lws_amd.lws(fsize, hop)
F = fsize // 2 + 1
rng = np.random.default_rng(5)
S = rng.standard_normal((1, T, F)) + 1j * rng.standard_normal((1, T, F))
thr = np.zeros(n_it)
np.set_printoptions(linewidth=250, precision=1)
for rows in ([0], [0, 1], [0, 7], [0, 2], [0, 3], [0, 4], [0, 5], [0, 6], list(range(8))):
    W = np.array(p.W).copy()
    for r in range(8):
        if r not in rows:
            W[:, r, :] = 0
    W[:, 0, 0] = p.W[:, 0, 0]
    pl = _capi.Plan(F, W)
    out = pl.batch(S, thr)
    name = pl.last_kernel()["name"]
    p64 = _capi.Plan(F, W, precision="fp64")
    ref = p64.batch(S[0], thr)
    d = np.abs(out[0] - ref)
    bad = d > 1e-3
    print(rows, name, "max err %.3g rel %.3g bad %d/%d magerr %.3g" % (d.max(), np.linalg.norm(out[0] - ref) / np.linalg.norm(ref), bad.sum(), bad.size, np.abs(np.abs(out[0]) - np.abs(S[0])).max()))
    if bad.sum() and len(sys.argv) > 3:
        for t in range(T):
            print("%3d " % t + "".join("X" if b else "." for b in bad[t]))
