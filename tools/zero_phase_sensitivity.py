import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, lws_amd
from oracle.oracle import Oracle
orc = Oracle()
rng = np.random.default_rng(2)
M = np.abs(rng.standard_normal((500, 513)) + 1j * rng.standard_normal((500, 513)))
thr = lws_amd.get_thresholds(100, 100, 0.1, 1)
S = M.astype(complex)
p = lws_amd.lws(1024, 256, precision="fp64")
ref = orc.batch_lws(S, p.W, thr)
# the oracle's own sensitivity: one ulp on one input bin
S2 = S.copy(); S2[250, 256] *= (1 + 2.3e-16)
ref2 = orc.batch_lws(S2, p.W, thr)
def stats(name, a, b):
    d = np.abs(a - b) / np.abs(b).max()
    print("%-34s median %.1e  99%% %.1e  99.9%% %.1e  max %.1e  bins > 1e-6: %d of %d" % (name, np.median(d), np.quantile(d, .99), np.quantile(d, .999), d.max(), (d > 1e-6).sum(), d.size))
stats("oracle, one ulp on one input bin", ref2, ref)
out = p.batch_lws(S); print(p.plan().last_kernel()["name"]); stats("fp64 systolic vs oracle", out, ref)
g = lws_amd.lws(1024, 256, precision="fp64", force_generic=True)
outg = g.batch_lws(S); print(g.plan().last_kernel()["name"]); stats("fp64 generic vs oracle", outg, ref)
stats("fp64 systolic vs generic", out, outg)
for it in (45, 50, 60, 80):
    t = thr[:it]
    r = orc.batch_lws(S, p.W, t)
    q = lws_amd.lws(1024, 256, precision="fp64", batch_iterations=it)
    o = q.batch_lws(S, thresholds=t)
    stats("systolic vs oracle after %d sweeps" % it, o, r)
