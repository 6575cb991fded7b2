import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, lws_amd
from conftest import load_golden
fp = load_golden("config2_fingerprint.npz")
rng = np.random.default_rng(int(fp["seed"]))
M = np.abs(rng.standard_normal((500, 513)) + 1j * rng.standard_normal((500, 513))).astype(np.float32).astype(np.float64)
for prec in ("fp64","fp32"):
    p = lws_amd.lws(1024, 256, precision=prec)
    t=time.time(); out_d = p.batch_lws(M, thresholds=np.zeros(20)); dt=time.time()-t
    d = np.abs(out_d.ravel()[::97] - fp["sample_dense20"])
    print(prec, 'dense20 rel', np.linalg.norm(d)/np.linalg.norm(fp["sample_dense20"]), 'median', np.median(d), 'cons', p.get_consistency(out_d), float(fp["consistency_dense20"]), 'time', dt, p.plan().last_kernel())
    t=time.time(); out = p.run_lws(M); dt=time.time()-t
    d = np.abs(out.ravel()[::97] - fp["sample_out"])
    print(prec, 'default100 rel', np.linalg.norm(d)/np.linalg.norm(fp["sample_out"]), 'median', np.median(d), 'q999', np.quantile(d,0.999), 'cons', p.get_consistency(out), float(fp["consistency_out"]), 'time', dt, p.plan().last_kernel())
