import sys, time, os; sys.path.insert(0,'.')
import numpy as np, lws_amd, torch
from bench import synth_magnitudes
def run(fs,sh,B,T,it,nwg=None):
    if nwg is None: os.environ.pop("LWS_SYSTOLIC_NWG",None)
    else: os.environ["LWS_SYSTOLIC_NWG"]=str(nwg)
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
    print(fs,sh,"B",B,"T",T,"it",it,"nwg",nwg,k, "%.3e bin-it/s"%(units/(k['ms']*1e-3)), flush=True)
    return state.cpu().numpy()
for (fs,sh,B,T,it) in [(1024,256,1,500,100),(1024,256,8,500,100),(1024,256,64,500,100),(1024,256,128,500,100),(2048,512,64,4000,60),(2048,512,16,4000,60)]:
    r1=run(fs,sh,B,T,it,1); rd=run(fs,sh,B,T,it)
    print("   identical:", np.array_equal(r1,rd))
