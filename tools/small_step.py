import os, sys
ROOT='/root/repo'
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd")): sys.path.insert(0,p)
import numpy as np, torch
from FBTT.tt_embeddings_ops import TTEmbeddingBag
n=2048
emb=TTEmbeddingBag(2449029,100,[16,16],[125,140,140],[4,5,5],sparse=True,use_cache=False,weight_dist="normal",learning_rate=0.01)
ids=torch.randperm(2449029)[:n].cuda(); offs=torch.arange(n+1).cuda(); d=torch.rand(n,100,device="cuda")
for _ in range(20): emb(ids,offs).backward(d)
torch.cuda.synchronize()
