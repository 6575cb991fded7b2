import sys, time, os; sys.path.insert(0,'.')
import numpy as np, lws_amd, torch
from bench import synth_magnitudes
B,T,F=256,500,513
p = lws_amd.lws(1024, 256); plan=p.plan()
mags = torch.from_numpy(synth_magnitudes(B, T, F, 1)).cuda()
state = torch.zeros((B,T,F),dtype=torch.complex64,device='cuda')
for it in (2,4,7,14,98,100):
    thr=np.zeros(it)
    for rep in range(3):
        state.copy_(mags); torch.cuda.synchronize()
        plan.batch_dev(state.data_ptr(), B, T, thr, stream=torch.cuda.current_stream().cuda_stream)
        k=plan.last_kernel()
    print(os.environ.get("LWS_HIP_LIB","default").split('/')[-1], "iters",it,"kernel ms %.3f"%k['ms'],flush=True)
