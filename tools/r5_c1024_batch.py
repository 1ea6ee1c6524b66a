"""N = 1024 complex float: us per launch against the batch (2^10 .. 2^20), for the launch shapes selected by the environment (development tool):
PFFFT_HIP_C1024_ONCE="W,rounds" (one transform per wavefront in dispatch order up to `rounds` resident sets; rounds 0 = always the loop),
PFFFT_HIP_C1024_WGS=<workgroups per CU of the loop>.  Prints one line per batch: us, fraction of 8 TB/s."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from r4_graph import per_call

tag = f"ONCE={os.environ.get('PFFFT_HIP_C1024_ONCE', 'default')} WGS={os.environ.get('PFFFT_HIP_C1024_WGS', '1')}"
for N, dt in ((1024, np.float32), (1024, np.float64)) if "--c5" in sys.argv else ((1024, np.float32),):
    s = pa.Setup(N, pa.COMPLEX, dt)
    row = []
    for lg in range(10, 21):
        B = 1 << lg
        x = torch.rand(B, s.vec_scalars, device="cuda", dtype=torch.float64 if dt == np.float64 else torch.float32)
        y = torch.empty_like(x)
        f = lambda: s.transform_batch(x, y, pa.FORWARD, True)
        for _ in range(5): f()
        n = max(20, min(2000, (1 << 31) // (x.numel() * x.element_size())))
        t = min(per_call(f, n) for _ in range(3))
        row.append(f"2^{lg}: {t:.1f} us ({2 * x.numel() * x.element_size() / t / 8e6:.3f})")
        del x, y
    print(f"[{tag}] N={N} {np.dtype(dt).name}: " + "  ".join(row), flush=True)
    s.close()
