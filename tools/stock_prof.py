import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
for N, o in ((2400, False), (4000, False), (800, True), (800, False)):
    batch = (1 << 30) // (N * 8)
    s = pa.Setup(N, pa.COMPLEX, np.float32)
    x = torch.rand(batch, 2 * N, device="cuda") * 2 - 1
    y = torch.empty_like(x)
    for _ in range(3): s.transform_batch(x, y, pa.FORWARD, o)
    torch.cuda.synchronize()
