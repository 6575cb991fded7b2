import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import lws_amd
B, T, F = 12, 200, 513
p = lws_amd.lws(1024, 256)
rng = np.random.default_rng(3)
M = np.abs(rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))).astype(np.complex128)
X = rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))
for name, S, thr in (("mag default", M, lws_amd.get_thresholds(100, 100, 0.1, 1)), ("complex alpha1", X, lws_amd.get_thresholds(30, 1, 0.1, 1)), ("mag dense", M, np.zeros(20))):
    plan = p.plan()
    os.environ.pop("LWS_HOST_CHUNK_BINS", None)
    a = plan.batch(S, thr)
    os.environ["LWS_HOST_CHUNK_BINS"] = "150000"
    b = plan.batch(S, thr)
    os.environ["LWS_HOST_MONOLITHIC"] = "1"
    c = plan.batch(S, thr)
    del os.environ["LWS_HOST_MONOLITHIC"]
    d = torch.from_numpy(S.astype(np.complex64)).cuda()
    plan.batch_dev(d.data_ptr(), B, T, thr); torch.cuda.synchronize()
    d = d.cpu().numpy()
    same_in = (a == S)
    print(name, "| chunked==unchunked", np.array_equal(a, b), "| vs monolithic max", np.abs(a - c).max(), "| vs dev max", np.abs(a - d).max(),
          "| untouched bins bit-identical to input:", int(same_in.sum()), "of", a.size, "| monolithic untouched:", int((c == S).sum()))
