import sys; sys.path.insert(0,'.')
import numpy as np, torch
from hdpgpc_amd import ops
from oracle import hdpgpc_oracle as orc
dev=lambda a: torch.as_tensor(a,dtype=torch.float64,device='cuda')
T,N,K=48,4,2
b=orc.synthetic_batch(N,K,T,seed=5); b["theta"][1,1]=2.5
plan=ops.PairsPlan(T,T,b["theta"]).update(dev(b["xb"]),dev(b["mean"]),dev(b["Sigma"]))
torch.cuda.synchronize()
print(plan.scalars().cpu().numpy()); print(plan.accuracy_bound())
quad,_,_=plan.loglik(dev(b["x"]),dev(b["y"]))
_,q_ref,_=orc.loglik_pairs(b["x"],b["y"],b["xb"],b["theta"],b["mean"],b["Sigma"])
print(np.abs(quad.cpu().numpy()-q_ref)/np.abs(q_ref))
