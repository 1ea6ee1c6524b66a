import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dt = np.float64 if "d" in sys.argv[2:] else np.float32
s = pa.Setup(N, pa.REAL, dt)
B = (1 << 30) // (N * np.dtype(dt).itemsize)
x = torch.rand(B, N, device="cuda", dtype=torch.float64 if dt == np.float64 else torch.float32)
y = torch.empty_like(x)
for var in (0, 121):
    pa.set_variant(var)
    for _ in range(30):
        s.transform_batch(x, y, pa.FORWARD, True)
    torch.cuda.synchronize()
pa.set_variant(0)
