import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from dma_ab import timed
logs = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "15,16,17,18,19,20,21,22,24").split(",")]
tag = f"PP={os.environ.get('PFFFT_HIP_TILE_PP','-')} PF={os.environ.get('PFFFT_HIP_TILE_PF','-')}"
for dtype in (np.float32, np.float64):
    row = []
    for lg in logs:
        N = 1 << lg
        s = pa.Setup(N, pa.COMPLEX, dtype)
        if pa.kernel_name(s) != "fourstep":
            s.close(); row.append(" n/a "); continue
        tdt = torch.float32 if dtype == np.float32 else torch.float64
        B = max(1, (1 << 30) // (s.vec_scalars * np.dtype(dtype).itemsize))
        x = torch.rand(B, s.vec_scalars, device="cuda", dtype=tdt) * 2 - 1
        y = torch.empty_like(x)
        t = min(timed(lambda: s.transform_batch(x, y, pa.FORWARD, True), 5) for _ in range(2))
        row.append(f"{2 * x.numel() * x.element_size() / t / 8e12:.3f}")
        del x, y; s.close(); torch.cuda.empty_cache()
    print(f"{tag} {np.dtype(dtype).name} cplx canonical 2^{logs}: " + " ".join(row), flush=True)
