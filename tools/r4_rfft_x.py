"""Real forward two sweeps (variant 122) against three (121), N = 2^16 .. 2^20, fraction of the roofline at 1 GiB (development tool)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from dma_ab import timed
for dt in (np.float32, np.float64):
    for lg in (16, 17, 18, 19, 20):
        N = 1 << lg
        s = pa.Setup(N, pa.REAL, dt)
        B = (1 << 30) // (N * np.dtype(dt).itemsize)
        x = torch.rand(B, N, device="cuda", dtype=torch.float64 if dt == np.float64 else torch.float32)
        y = torch.empty_like(x)
        row = []
        outs = {}
        for var in (121, 122):
            pa.set_variant(var)
            for o in (True, False):
                t = min(timed(lambda: s.transform_batch(x, y, pa.FORWARD, o), 10) for _ in range(2))
                row.append(f"v{var} {'ord' if o else 'unord'} {2 * x.numel() * x.element_size() / t / 8e12:.3f}")
            fo = s.transform_batch(x[:3], None, pa.FORWARD, True); fu = s.transform_batch(x[:3], None, pa.FORWARD, False)
            row.append("bit-identical" if torch.equal(s.zreorder_batch(fu, None, pa.FORWARD), fo) else "NOT IDENTICAL")
        pa.set_variant(0)
        print(np.dtype(dt).name, "N=2^%d" % lg, "  ".join(row), flush=True)
        s.close(); del x, y
