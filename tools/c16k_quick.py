import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from dma_ab import timed, relerr
from oracle import ref as oref
R = oref.get()
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,83,84,85").split(",")]
for N, tr in ((16384, pa.COMPLEX), (32768, pa.REAL)):
    s = pa.Setup(N, tr)
    rs = R.setup(N, tr)
    B = (1 << 32) // (s.vec_scalars * 4)
    x = torch.rand(B, s.vec_scalars, device="cuda") * 2 - 1
    y = torch.empty_like(x)
    xh = x[:3].cpu().numpy()
    for d in (pa.FORWARD, pa.BACKWARD):
        for o in (False, True):
            row = []
            want = rs.batch(xh, d, o)
            for var in variants:
                pa.set_variant(var)
                t = min(timed(lambda: s.transform_batch(x, y, d, o), 8) for _ in range(2))
                e = relerr(y[:3].cpu().numpy(), want)
                row.append(f"v{var}: {2 * x.numel() * 4 / t / 8e12:.3f} ({e:.0e})")
            pa.set_variant(0)
            print(f"N={N:6d} {'cplx' if tr else 'real'} {'fwd' if d == 0 else 'bwd'} {'canonical' if o else 'internal '}: " + "  ".join(row), flush=True)
    del x, y; s.close(); rs.close()
