"""Per-pass kernel times of complex transforms beyond LDS (development tool, run under tools/kstats.sh): N1,N2,... [d]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
dt = np.float64 if "d" in sys.argv[2:] else np.float32
for N in [int(v) for v in sys.argv[1].split(",")]:
    s = pa.Setup(N, pa.COMPLEX, dt)
    isz = np.dtype(dt).itemsize
    B = (1 << 29) // (2 * N * isz)
    x = torch.rand(B, 2 * N, device="cuda", dtype=torch.float64 if dt == np.float64 else torch.float32)
    y = torch.empty_like(x)
    for _ in range(20):
        s.transform_batch(x, y, pa.FORWARD, True)
    torch.cuda.synchronize()
    print(N, B, 2 * x.numel() * isz, "bytes per pass at the roofline:", 2 * x.numel() * isz / 8e12 * 1e6, "us")
    s.close()
    del x, y
