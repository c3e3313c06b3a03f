import numpy as np, torch, sys
sys.path.insert(0,".")
from secp256k1_zkp_amd import Engine
from tests.refapi import G_XY
e=Engine(0); rng=np.random.default_rng(1)
n=1<<21
a=torch.tensor(np.frombuffer(G_XY,np.uint8).copy()).cuda().repeat(n,1)
na=torch.tensor(rng.integers(0,256,(n,32),dtype=np.uint8)).cuda(); ng=torch.tensor(rng.integers(0,256,(n,32),dtype=np.uint8)).cuda()
r=torch.zeros(n*64,dtype=torch.uint8,device="cuda"); inf=torch.zeros(n,dtype=torch.int32,device="cuda")
for it in range(3):
    e.ecmult_batch_dev(r,inf,a,na,ng); e.sync(); print("ecmult_batch n=2^21: %.3f ms  -> %.3e ecmult/s"%(e.last_ms(1),n/e.last_ms(1)*1e3))
