import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
print("cpu_count", os.cpu_count(), "torch threads", torch.get_num_threads(), flush=True)
from oracle import lavender_ref as R
t0=time.time(); P = R.filled_params("base"); print("fill", time.time()-t0, flush=True)
for thr in (8, 16, 32):
    torch.set_num_threads(thr)
    g = torch.Generator().manual_seed(0)
    batch = dict(img=torch.randn(2,5,3,224,224,generator=g), txt=torch.randint(1000,30000,(2,32),generator=g), mask=torch.ones(2,32,dtype=torch.long), ans_mtm=torch.randint(0,30000,(2,32),generator=g))
    for it in range(2):
        t0=time.time()
        with torch.no_grad():
            np.random.seed(0); out = R.pretrain_forward(P, batch, "base", 12)
        print("threads", thr, "fwd iter", it, round(time.time()-t0,2), flush=True)
