import sys; sys.path.insert(0,'.')
import numpy as np, torch
from hdpgpc_amd import ops
from oracle import hdpgpc_oracle as orc
dev=lambda a: torch.as_tensor(a,dtype=torch.float64,device='cuda')
for T in (144, 192, 256):
    rng=np.random.default_rng(T)
    Q=rng.normal(size=(2,T,T)); A=Q@Q.transpose(0,2,1)/T+np.eye(T)
    Y=rng.normal(size=(5,T)); mean=rng.normal(size=(2,T))
    im,ia,io,ic=ops.build_items([0,1],[0.0,0.0],[3,2])
    quad,ld,info=ops.score_groups(dev(Y),dev(mean),dev(A),im,ia,io,ic,jitter_rel=0.0,want_logdet=True)
    for n in range(5):
        k=0 if n<3 else 1
        q,l=orc.quad_logdet(Y[n]-mean[k],A[k]) if False else (None,None)
        L=np.linalg.cholesky(A[k]); z=np.linalg.solve(L,Y[n]-mean[k]); q=z@z; l=2*np.log(np.diag(L)).sum()
        print(T,n,"quad rel err",abs(float(quad[n])-q)/q,"logdet err",abs(float(ld[n])-l), int(info[n]))
