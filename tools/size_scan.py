"""Scan of legal sizes for slow outliers (development tool): forward canonical + unordered, 256 MiB per launch, 10 + 10 launches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pffft_amd as pa

def smooth5(m):
    for q in (2, 3, 5):
        while m % q == 0: m //= q
    return m == 1

lo, hi = int(sys.argv[1]), int(sys.argv[2])
dt = np.float64 if len(sys.argv) > 3 and sys.argv[3] == "f64" else np.float32
tdt = torch.float64 if dt == np.float64 else torch.float32
step = 16
sizes = [n for n in range(max(16, lo), hi + 1, step) if smooth5(n)]
if len(sizes) > 160: sizes = sizes[:: len(sizes) // 160 + 1]
for tr, name in ((pa.COMPLEX, "cplx"), (pa.REAL, "real")):
    for n in sizes:
        N = n if tr == pa.COMPLEX else 2 * n
        try: s = pa.Setup(N, tr, dt)
        except Exception as e: print(name, N, "setup failed", e); continue
        batch = max(1, (1 << 28) // (s.vec_scalars * np.dtype(dt).itemsize))
        x = torch.rand(batch, s.vec_scalars, device="cuda", dtype=tdt) * 2 - 1
        y = torch.empty_like(x)
        res = []
        for o in (True, False):
            f = lambda: s.transform_batch(x, y, pa.FORWARD, o)
            for _ in range(10): f()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): f()
            b.record(); torch.cuda.synchronize()
            res.append(2 * x.numel() * x.element_size() / (a.elapsed_time(b) / 10 * 1e-3) / 8e12)
        print(f"{name} N={N:7d} [{pa.kernel_name(s):9s}] ordered {res[0]:.3f} unordered {res[1]:.3f}", flush=True)
        del x, y; torch.cuda.empty_cache(); s.close()
