import sys, time; sys.path.insert(0,'.')
import numpy as np, lws_amd, torch
from bench import synth_magnitudes
for (fs,sh,B,T,it) in [(2048,512,128,500,100),(2048,512,64,4000,50),(1024,256,256,500,100)]:
    F=fs//2+1
    p = lws_amd.lws(fs, sh); plan=p.plan()
    mags = torch.from_numpy(synth_magnitudes(B, T, F, 1)).cuda()
    state = torch.zeros((B,T,F),dtype=torch.complex64,device='cuda')
    thr=np.zeros(it)
    for rep in range(2):
        state.copy_(mags); torch.cuda.synchronize(); t=time.time()
        plan.batch_dev(state.data_ptr(), B, T, thr, stream=torch.cuda.current_stream().cuda_stream)
        k=plan.last_kernel(); torch.cuda.synchronize(); dt=time.time()-t
    units=B*T*F*it
    print(fs,sh,B,T,it,k, "wall %.1f ms  %.3e bin-it/s  alg %.0f GB/s (%.1f%% of 8TB/s)"%(dt*1e3, units/(k['ms']*1e-3), 20*units/(k['ms']*1e-3)/1e9, 20*units/(k['ms']*1e-3)/8e12*100), flush=True)
