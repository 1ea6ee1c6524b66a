"""Representative Stockham-kernel launches for rocprofv3 (tools/profile_cmd.sh stock tools/stock_prof.py)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
CASES = [(96, 1, "f"), (480, 1, "f"), (800, 1, "f"), (2400, 1, "f"), (4000, 1, "f"), (9216, 1, "f"),
         (480, 0, "f"), (2400, 0, "f"), (9216, 0, "f"), (16384, 0, "f"), (480, 1, "d"), (2400, 1, "d"), (4096, 1, "d")]
for N, tr, pr in CASES:
    dt = np.float32 if pr == "f" else np.float64
    s = pa.Setup(N, tr, dt)
    batch = (1 << 30) // (s.vec_scalars * np.dtype(dt).itemsize)
    x = torch.rand(batch, s.vec_scalars, device="cuda", dtype=torch.float32 if pr == "f" else torch.float64) * 2 - 1
    y = torch.empty_like(x)
    for _ in range(3): s.transform_batch(x, y, pa.FORWARD, False)
    torch.cuda.synchronize()
    print(N, tr, pr, batch, flush=True)
    del x, y
