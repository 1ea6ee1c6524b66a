"""Launch-fixed cost of the LDS-resident kernels: time per call against the batch (development tool).  N,transform,dtype triples."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from r4_graph import per_call
VAR = int(os.environ.get("R4_VARIANT", "0"))
pa.set_variant(VAR)
SIZES = os.environ.get('R4_SIZES')
DEFAULT = ((1024, pa.COMPLEX, np.float32), (1024, pa.COMPLEX, np.float64), (16384, pa.REAL, np.float32), (4096, pa.COMPLEX, np.float32), (2048, pa.COMPLEX, np.float32), (256, pa.COMPLEX, np.float32), (512, pa.REAL, np.float32), (2048, pa.COMPLEX, np.float64))
if SIZES:
    DEFAULT = tuple((int(a.split(':')[0]), pa.REAL if a.split(':')[1] == 'r' else pa.COMPLEX, np.float64 if a.split(':')[2] == 'd' else np.float32) for a in SIZES.split(','))
for N, tr, dt in DEFAULT:
    s = pa.Setup(N, tr, dt)
    vb = s.vec_scalars * np.dtype(dt).itemsize
    row = []
    for mib in (2, 8, 32, 64, 128, 256, 512, 1024):
        B = max(1, (mib << 20) // vb)
        x = torch.rand(B, s.vec_scalars, device="cuda", dtype=torch.float64 if dt == np.float64 else torch.float32); y = torch.empty_like(x)
        s.transform_batch(x, y, pa.FORWARD, False)
        t = min(per_call(lambda: s.transform_batch(x, y, pa.FORWARD, False), max(20, 2048 // mib)) for _ in range(3))
        row.append(f"{mib} MiB: {t:.1f} us ({2 * B * vb / t / 8e6:.2f})")
        del x, y
    print(f"N={N} {'cplx' if tr == pa.COMPLEX else 'real'} {np.dtype(dt).name} [{pa.kernel_name(s)}]: " + "  ".join(row), flush=True)
    s.close()
