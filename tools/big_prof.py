import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
for lg in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "16,18,20,22").split(",")]:
    N = 1 << lg
    s = pa.Setup(N, pa.COMPLEX, np.float32)
    B = (1 << 30) // (8 * N)
    x = torch.rand(B, 2 * N, device="cuda") * 2 - 1
    y = torch.empty_like(x)
    for _ in range(5):
        s.transform_batch(x, y, pa.FORWARD, True)
    torch.cuda.synchronize()
    s.close()
