"""Throughput of the elementwise spectrum helpers (zreorder, zconvolve) on device-resident batches."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from bench_configs import timed
for N, tr, dt in ((1024, pa.COMPLEX, np.float32), (16384, pa.REAL, np.float32), (96, pa.COMPLEX, np.float32), (1024, pa.COMPLEX, np.float64)):
  for var in [int(v) for v in os.environ.get("PFVS", "0,60").split(",")]:
    pa.set_variant(var); print("variant", var)
    s = pa.Setup(N, tr, dt)
    tdt = torch.float32 if dt == np.float32 else torch.float64
    batch = (1 << 30) // (s.vec_scalars * np.dtype(dt).itemsize)
    x = torch.rand(batch, s.vec_scalars, device="cuda", dtype=tdt)
    y = torch.empty_like(x); z = torch.empty_like(x)
    byts = batch * s.vec_scalars * x.element_size()
    for d, nm in ((pa.FORWARD, "zreorder fwd"), (pa.BACKWARD, "zreorder bwd")):
        t = timed(lambda: s.zreorder_batch(x, y, d))
        print(f"N={N} tr={tr} {dt.__name__} {nm}: {t*1e3:.3f} ms {2*byts/t/1e9:.0f} GB/s frac={2*byts/t/8e12:.3f}")
    t = timed(lambda: s.zconvolve_batch(x, y, z, 1.0, accumulate=False))
    print(f"N={N} tr={tr} {dt.__name__} zconvolve no_accu: {t*1e3:.3f} ms {3*byts/t/1e9:.0f} GB/s frac={3*byts/t/8e12:.3f}")
    t = timed(lambda: s.zconvolve_batch(x, y, z, 1.0, accumulate=True))
    print(f"N={N} tr={tr} {dt.__name__} zconvolve accumulate: {t*1e3:.3f} ms {4*byts/t/1e9:.0f} GB/s frac={4*byts/t/8e12:.3f}")
    t = timed(lambda: z.copy_(x))
    print(f"N={N} torch copy: {2*byts/t/1e9:.0f} GB/s")
    s.close()
