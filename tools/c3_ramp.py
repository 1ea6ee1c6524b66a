import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
mode = sys.argv[1] if len(sys.argv) > 1 else "empty"
s = pa.Setup(16384, pa.REAL)
B = 1 << 16
if mode == "c2first":
    s2 = pa.Setup(1024, pa.COMPLEX)
    x2 = torch.empty(1 << 20, 2048, device="cuda").uniform_(-1, 1); y2 = torch.empty_like(x2)
    for _ in range(13): s2.transform_batch(x2, y2, pa.FORWARD, False)
    torch.cuda.synchronize(); del x2, y2; torch.cuda.empty_cache()
x = torch.empty(B, 16384, device="cuda").uniform_(-1, 1)
y = torch.empty_like(x) if mode != "zeros" else torch.zeros_like(x)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
ev[0].record()
for i in range(60):
    s.transform_batch(x, y, pa.FORWARD, False)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(60)]
fr = [2 * x.numel() * 4 / (t * 1e-3) / 8e12 for t in ts]
print(mode, " ".join(f"{f:.3f}" for f in fr[:16]), "... last10 avg", round(sum(fr[-10:]) / 10, 3), flush=True)
