import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
L, taps = 1 << 20, 4096
x = torch.rand(L, device="cuda") * 2 - 1
h = np.random.default_rng(0).uniform(-1, 1, taps).astype(np.float32)
fc = pa.FastConv(h, 0, 0)
y = torch.empty_like(x)
for v in (0, 79, 30):
    pa.set_variant(v)
    for _ in range(5): fc.apply(x, True, out=y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): fc.apply(x, True, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): fc.apply(x, True, out=y)
    b.record(); torch.cuda.synchronize()
    print("variant", v, "wall per call us", (time.perf_counter() - t0) / 400 * 1e6, "gpu us per call", a.elapsed_time(b) * 1e3 / 200, "checksum", float(y[:1 << 19].double().sum()))
pa.set_variant(0)
