import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
for N, tr in ((1200, pa.COMPLEX), (2400, pa.REAL), (4608, pa.COMPLEX), (9216, pa.REAL), (800, pa.COMPLEX), (768, pa.COMPLEX)):
    s = pa.Setup(N, tr, np.float32)
    B = (1 << 30) // (s.vec_scalars * 4)
    x = torch.rand(B, s.vec_scalars, device="cuda"); y = torch.empty_like(x)
    for o in (True, False):
        for _ in range(12):
            s.transform_batch(x, y, pa.FORWARD, o)
    torch.cuda.synchronize(); s.close(); del x, y
