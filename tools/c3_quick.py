"""C3 (N=16384 real float, batch 2^16) and N=8192 complex: fraction of 8 TB/s for a list of kernel variants."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from dma_ab import timed
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,79,96,98").split(",")]
for N, tr in ((16384, pa.REAL), (8192, pa.COMPLEX)):
    s = pa.Setup(N, tr)
    B = (1 << 32) // (s.vec_scalars * 4)
    x = torch.rand(B, s.vec_scalars, device="cuda") * 2 - 1
    y = torch.empty_like(x)
    for d in (pa.FORWARD, pa.BACKWARD):
        for o in (False, True):
            row = []
            for var in variants:
                pa.set_variant(var)
                t = min(timed(lambda: s.transform_batch(x, y, d, o), 10) for _ in range(2))
                row.append(f"v{var}: {2 * x.numel() * 4 / t / 8e12:.3f}")
            pa.set_variant(0)
            print(f"N={N:6d} {'cplx' if tr else 'real'} {'fwd' if d == 0 else 'bwd'} {'canonical' if o else 'internal '}: " + "  ".join(row), flush=True)
    del x, y; s.close()
