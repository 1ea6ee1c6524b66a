"""Fraction of 8 TB/s (2 x vector bytes, 1 GiB per launch) and parity against oracle/_ref for a list of sizes: N[:r|c][:d]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from dma_ab import timed
from oracle import ref as oref
R = oref.get()
for spec in sys.argv[1:]:
    parts = spec.split(":")
    N = int(parts[0]); tr = pa.REAL if "r" in parts[1:] else pa.COMPLEX; dt = np.float64 if "d" in parts[1:] else np.float32
    tdt = torch.float64 if dt == np.float64 else torch.float32
    s = pa.Setup(N, tr, dt); rs = R.setup(N, tr, dt)
    B = max(8, (1 << 30) // (s.vec_scalars * np.dtype(dt).itemsize))
    x = torch.rand(B, s.vec_scalars, device="cuda", dtype=tdt) * 2 - 1
    y = torch.empty_like(x)
    row, worst = [], 0.0
    timed(lambda: s.transform_batch(x, y, pa.FORWARD, True), 1, warm=40)
    for d in (pa.FORWARD, pa.BACKWARD):
        for o in (True, False):
            t = min(timed(lambda: s.transform_batch(x, y, d, o), 20) for _ in range(2))
            row.append(f"{2 * x.numel() * x.element_size() / t / 8e12:.3f}")
            for i in (0, B // 2, B - 1):
                want = (rs.transform_ordered if o else rs.transform_unordered)(x[i].cpu().numpy(), d)
                worst = max(worst, float(np.abs(y[i].cpu().numpy() - want).max() / np.abs(want).max()))
    print(f"N={N:8d} {'real' if tr == pa.REAL else 'cplx'} {np.dtype(dt).name}: {pa.kernel_name(s):9s} [fwd ord, fwd unord, bwd ord, bwd unord] {' '.join(row)}  worst rel err {worst:.2e}", flush=True)
    s.close(); rs.close(); del x, y
