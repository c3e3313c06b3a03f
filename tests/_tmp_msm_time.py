import numpy as np, torch, sys, time
sys.path.insert(0,".")
from secp256k1_zkp_amd import Engine
from tests.refapi import Ref, G_XY
ref=Ref(); e=Engine(0); rng=np.random.default_rng(1)
for lg in (14, 17, 20):
    n=1<<lg
    k=rng.integers(0,256,(n,32),dtype=np.uint8); g=np.frombuffer(G_XY*n,np.uint8).reshape(n,64)
    pts,_=e.ecmult_batch(g,np.zeros((n,32),np.uint8),k)
    sc=rng.integers(0,256,(n,32),dtype=np.uint8)
    dsc=torch.tensor(sc).cuda(); dpt=torch.tensor(pts).cuda()
    r=torch.zeros(64,dtype=torch.uint8,device="cuda"); ri=torch.zeros(1,dtype=torch.int32,device="cuda")
    for it in range(3):
        e.ecmult_multi_dev(r,ri,dsc,dpt); e.sync()
        print("msm n=2^%d: total %.2f ms (bucket kernel %.2f ms) -> %.1f Mpoint-scalar/s"%(lg,e.last_ms(0),e.last_ms(1),n/e.last_ms(0)/1e3))
    if lg<=17:
        t=time.time(); exp,einf=ref.ecmult_multi(sc,pts,None,None); dt=time.time()-t
        print("   ref %.2fs (%.3f Mpt/s) match=%s"%(dt,n/dt/1e6,np.array_equal(exp,r.cpu().numpy())))
