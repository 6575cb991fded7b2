import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, lws_amd
from oracle.oracle import Oracle
o=Oracle()
p=lws_amd.lws(64,16)
rng=np.random.default_rng(0)
# all-zero spectrogram
Z=np.zeros((12,33),complex)
for fn in (lambda S: p.batch_lws(S, thresholds=[0.0,0.0]), lambda S: lws_amd.lws(64,16,mode='music').run_lws(S)):
    out=fn(Z); print("zeros ->", np.abs(out).max(), np.isnan(out).any())
# zeros in some bins + negative thresholds
S=rng.standard_normal((20,33))+1j*rng.standard_normal((20,33)); S[3:6,:]=0; S[:,7]=0
thr=[-1.0,0.0,0.5]
out=p.batch_lws(S,thresholds=thr); ref=o.batch_lws(S,p.W,thr)
print("zeros/neg thr: rel", np.linalg.norm(out-ref)/np.linalg.norm(ref), "nan", np.isnan(out).any())
# > 440 iterations -> generic
thr=np.full(450,0.3); out=p.batch_lws(S,thresholds=thr); print("450 iters kernel", p.plan().last_kernel()['name'], np.isnan(out).any())
ref=o.batch_lws(S,p.W,thr); print("  rel", np.linalg.norm(out-ref)/np.linalg.norm(ref))
# B=0 / T=0
try:
    print("B=0", lws_amd._capi.Plan(33,p.W).batch(np.zeros((0,5,33),complex),[0.0]).shape)
except Exception as e: print("B=0 exc", type(e).__name__, e)
# huge/small scale
for sc in (1e-20,1e20):
    out=p.batch_lws(S*sc,thresholds=[0.0,0.0,0.0]); ref=o.batch_lws(S*sc,p.W,[0.0,0.0,0.0])
    print("scale",sc,"rel",np.linalg.norm(out-ref)/np.linalg.norm(ref), np.isnan(out).any())
# inf / nan input
S2=S.copy(); S2[5,5]=np.nan
out=p.batch_lws(S2,thresholds=[0.0]); print("nan input -> nan count", np.isnan(out).sum(), "of", out.size)
