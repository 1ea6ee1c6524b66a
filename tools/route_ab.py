"""A/B of the kernel families on the power-of-two sizes both serve (development tool): default routing vs variant 50 (Stockham
compile-time plans wherever they exist) vs variant 54 (always the register-tiled kernel), 1 GiB per launch, 10 + 20 launches,
[fwd ordered, fwd unordered, bwd ordered, bwd unordered] as fractions of 8 TB/s."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa

def rate(s, x, y, d, o):
    f = lambda: s.transform_batch(x, y, d, o)
    for _ in range(10): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    return 2 * x.numel() * x.element_size() / (a.elapsed_time(b) / 20 * 1e-3) / 8e12

for dt, tdt in ((np.float32, torch.float32), (np.float64, torch.float64)):
    pool = torch.rand((1 << 30) // np.dtype(dt).itemsize, device="cuda", dtype=tdt) * 2 - 1
    ypool = torch.empty_like(pool)
    for tr, name in ((pa.COMPLEX, "cplx"), (pa.REAL, "real")):
        for n in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
            N = n if tr == pa.COMPLEX else 2 * n
            row = []
            for v in (0, 50, 54):
                pa.set_variant(v)
                try:
                    s = pa.Setup(N, tr, dt)
                    b = pool.numel() // s.vec_scalars
                    x = pool[: b * s.vec_scalars].view(b, -1); y = ypool[: b * s.vec_scalars].view(b, -1)
                    r = [rate(s, x, y, d, o) for d in (pa.FORWARD, pa.BACKWARD) for o in (True, False)]
                    row.append(" ".join(f"{q:.3f}" for q in r))
                    s.close()
                except Exception as e:
                    row.append("err " + str(e)[:40])
                pa.set_variant(0)
            print(f"{np.dtype(dt).name} {name} N={N:6d}  default [{row[0]}]  stockham(50) [{row[1]}]  tiled(54) [{row[2]}]", flush=True)
    del pool, ypool; torch.cuda.empty_cache()
