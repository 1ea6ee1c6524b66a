"""Sensitivity of the Stockham kernels to the launch knobs (env, one process per setting): fraction of 8 TB/s per size, four combos."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from dma_ab import timed
specs = sys.argv[1:]
tag = " ".join(f"{k[10:]}={v}" for k, v in os.environ.items() if k.startswith("PFFFT_HIP_"))
for spec in specs:
    parts = spec.split(":")
    N = int(parts[0]); tr = pa.REAL if "r" in parts[1:] else pa.COMPLEX; dt = np.float64 if "d" in parts[1:] else np.float32
    s = pa.Setup(N, tr, dt)
    B = (1 << 30) // (s.vec_scalars * np.dtype(dt).itemsize)
    x = torch.rand(B, s.vec_scalars, device="cuda", dtype=torch.float64 if dt == np.float64 else torch.float32)
    y = torch.empty_like(x)
    timed(lambda: s.transform_batch(x, y, pa.FORWARD, True), 1, warm=40)
    row = []
    for d in (pa.FORWARD, pa.BACKWARD):
        for o in (True, False):
            t = min(timed(lambda: s.transform_batch(x, y, d, o), 20) for _ in range(2))
            row.append(f"{2 * x.numel() * x.element_size() / t / 8e12:.3f}")
    print(f"[{tag}] {spec:10s} {' '.join(row)}", flush=True)
    s.close(); del x, y
