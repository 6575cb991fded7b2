import sys, time; sys.path.insert(0,'.')
import numpy as np, torch, lws_amd
for fs,hop in ((768,256),(1024,384),(1020,170),(1000,200),(1024,224)):
    p=lws_amd.lws(fs,hop); F=fs//2+1
    x=torch.rand((256,500,F),device='cuda').to(torch.complex64)
    thr=np.zeros(40)
    for _ in range(2):
        p.plan().batch_dev(x.data_ptr(),256,500,thr,stream=torch.cuda.current_stream().cuda_stream); k=p.plan().last_kernel()
    print(fs,hop,k['name'],round(k['ms'],2),'ms for 40 sweeps ->', round(k['ms']*1e9/(256*500*F*40),2),'ps/bin-sweep')
