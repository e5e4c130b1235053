import time, torch, numpy as np
print("threads", torch.get_num_threads(), torch.get_num_interop_threads())
N,K=24,24
resp=torch.zeros((N,K),dtype=torch.float64); resp[torch.arange(N),torch.arange(N)%K]=1
pair=torch.zeros((N,K,K),dtype=torch.float64)
def f():
    order=torch.argsort(torch.sum(resp,dim=0),descending=True)
    r=resp[:,order]; p=pair[:,order,:][:,:,order]
    return order
for nt in (None,1):
    if nt: torch.set_num_threads(nt)
    for rep in range(3):
        t0=time.perf_counter()
        for _ in range(200): f()
        print(nt,"reorder host ops us:",(time.perf_counter()-t0)/200*1e6)
x=torch.zeros(10,device='cuda'); torch.cuda.synchronize()
o=torch.arange(24)
t0=time.perf_counter()
for _ in range(200): od=o.to('cuda')
torch.cuda.synchronize(); print("H2D .to us", (time.perf_counter()-t0)/200*1e6)
q=torch.zeros((24,24,1),device='cuda',dtype=torch.float64)
t0=time.perf_counter()
for _ in range(200): qq=q[:,od]
torch.cuda.synchronize(); print("gpu index us", (time.perf_counter()-t0)/200*1e6)
a=np.random.rand(60)
t0=time.perf_counter()
for _ in range(20000): np.sum(a)
print("np.sum us",(time.perf_counter()-t0)/20000*1e6)
