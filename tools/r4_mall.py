"""Does the work buffer of the two-pass route stay in the memory-side cache when the batch is small?  Fraction of the roofline of complex
float N = 2^18 / 2^16 / 2^20 over batch sizes from 16 MiB to 1 GiB, out of place and in place (development tool)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from dma_ab import timed
for N in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "262144,65536,1048576").split(",")]:
    s = pa.Setup(N, pa.COMPLEX)
    for mib in (16, 32, 64, 128, 256, 512, 1024):
        B = max(1, (mib << 20) // (8 * N))
        x = torch.rand(B, 2 * N, device="cuda") * 2 - 1
        y = torch.empty_like(x)
        row = []
        for inplace in (0, 1):
            f = (lambda: s.transform_batch(x, x, pa.FORWARD, True)) if inplace else (lambda: s.transform_batch(x, y, pa.FORWARD, True))
            reps = max(10, 4096 // mib)
            t = min(timed(f, reps) for _ in range(3))
            row.append(f"{2 * x.numel() * 4 / t / 8e12:.3f} ({t * 1e6:.0f} us)")
        print(f"N={N} batch {B:5d} = {mib:5d} MiB: out of place {row[0]}   in place {row[1]}", flush=True)
        del x, y
    s.close()
