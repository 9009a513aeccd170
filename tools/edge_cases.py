import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/falcon-ttdforgnns_amd')
import numpy as np, torch
from FBTT.tt_embeddings_ops import TTEmbeddingBag
from oracle import tt_oracle as orc
p,q,r=[20,25,30],[4,5,5],[16,16]
n=int(np.prod(p))
emb=TTEmbeddingBag(n,100,r,p,q,sparse=False,use_cache=False,weight_dist="normal")
cores=[c.detach()[0].cpu().numpy() for c in emb.tt_cores]
def check(name, idx, offs):
    out=emb(idx,offs)
    want=orc.tt_forward(idx.cpu().numpy().astype(np.int64), offs.cpu().numpy().astype(np.int64), cores,p,q,[1]+r+[1])
    err=float(np.abs(out.detach().cpu().numpy()-want).max()) if want.size else 0.0
    print(name, tuple(out.shape), f"{err:.2e}")
    return out
ids=torch.randint(0,n,(5000,))
check("int64", ids.cuda(), torch.arange(5001).cuda())
check("int32 ids", ids.int().cuda(), torch.arange(5001).int().cuda())
nc=torch.stack([ids,ids],1).cuda()[:,0]
print("noncontig", nc.is_contiguous()); check("noncontig ids", nc, torch.arange(5001).cuda())
check("empty", torch.zeros(0,dtype=torch.int64).cuda(), torch.zeros(1,dtype=torch.int64).cuda())
check("empty bags only", torch.zeros(0,dtype=torch.int64).cuda(), torch.zeros(8,dtype=torch.int64).cuda())
o=check("big bag", ids.cuda(), torch.tensor([0,5000]).cuda())
o.sum().backward(); print("grad ok", all(torch.isfinite(c.grad).all().item() for c in emb.tt_cores))
try:
    emb(ids, torch.arange(5001))
except RuntimeError as e:
    print("cpu tensor:", str(e)[:80])
# requires_grad on output used in a bigger graph
emb.zero_grad()
w=torch.randn(100,7,device="cuda",requires_grad=True)
loss=(emb(ids.cuda(), torch.arange(5001).cuda()) @ w).pow(2).mean(); loss.backward()
print("composed grads", w.grad.abs().sum().item()>0, emb.tt_cores[1].grad.abs().sum().item()>0)
# eval / no_grad
with torch.no_grad():
    o=emb(ids.cuda(), torch.arange(5001).cuda()); print("no_grad", o.requires_grad)
