"""A/B of two variants on a list of sizes (N[:r][:d]), alternating runs: fraction of 8 TB/s, four combos each, best of 3."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from dma_ab import timed
va, vb = int(sys.argv[1]), int(sys.argv[2])
for spec in sys.argv[3:]:
    parts = spec.split(":")
    N = int(parts[0]); tr = pa.REAL if "r" in parts[1:] else pa.COMPLEX; dt = np.float64 if "d" in parts[1:] else np.float32
    s = pa.Setup(N, tr, dt)
    B = (1 << 30) // (s.vec_scalars * np.dtype(dt).itemsize)
    x = torch.rand(B, s.vec_scalars, device="cuda", dtype=torch.float64 if dt == np.float64 else torch.float32)
    y = torch.empty_like(x)
    timed(lambda: s.transform_batch(x, y, pa.FORWARD, True), 1, warm=40)
    res = {va: [0.0] * 4, vb: [0.0] * 4}
    for rep in range(3):
        for v in (va, vb):
            pa.set_variant(v)
            i = 0
            for d in (pa.FORWARD, pa.BACKWARD):
                for o in (True, False):
                    t = timed(lambda: s.transform_batch(x, y, d, o), 20)
                    res[v][i] = max(res[v][i], 2 * x.numel() * x.element_size() / t / 8e12); i += 1
    pa.set_variant(0)
    print(f"{spec:10s} v{va}: {' '.join('%.3f' % r for r in res[va])}   v{vb}: {' '.join('%.3f' % r for r in res[vb])}", flush=True)
    s.close(); del x, y
