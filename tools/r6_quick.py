"""Quick fractions of the roofline for a list of (N, real|complex, f32|f64) at 1 GiB per launch, four direction x layout combinations,
min of 3 x 10 launches (development tool: A/Bs of one build against another on the same box)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa

def frac(N, tr, dtype):
    s = pa.Setup(N, tr, dtype)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    batch = max(1, (1 << 30) // (s.vec_scalars * np.dtype(dtype).itemsize))
    x = torch.rand((batch, s.vec_scalars), device="cuda", dtype=tdt) * 2 - 1
    y = torch.empty_like(x)
    out = []
    for d in (pa.FORWARD, pa.BACKWARD):
        for o in (True, False):
            f = lambda: s.transform_batch(x, y, d, o)
            for _ in range(5): f()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10): f()
                b.record(); torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b) / 10 * 1e-3)
            out.append(2 * x.numel() * x.element_size() / best / 8e12)
    s.close()
    return out

for spec in sys.argv[1:]:
    N, tr, dt = spec.split(":")
    r = frac(int(N), pa.REAL if tr == "r" else pa.COMPLEX, np.float32 if dt == "f32" else np.float64)
    print(f"{spec:>16}: fwd ord {r[0]:.3f}  fwd uno {r[1]:.3f}  bwd ord {r[2]:.3f}  bwd uno {r[3]:.3f}", flush=True)
