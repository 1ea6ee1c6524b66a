import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
L, taps = 1 << 20, 4096
x = torch.rand(L, device="cuda") * 2 - 1
h = np.random.default_rng(0).uniform(-1, 1, taps).astype(np.float32)
fc = pa.FastConv(h, 0, 0)
y = torch.empty_like(x)
for v in (0, 30):
    pa.set_variant(v)
    for _ in range(5): fc.apply(x, True, out=y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): fc.apply(x, True, out=y)
    torch.cuda.synchronize()
    print("variant", v, "wall per call us", (time.perf_counter() - t0) / 200 * 1e6)
pa.set_variant(0)
