"""Steady-state throughput of the Stockham kernels on non-power-of-two sizes (development tool): 20 untimed launches,
then 20 timed ones, 512 MiB of vectors per launch.  STOCK_AB_LIB=<path> measures another build of the library (A/B)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pffft_amd as pa
from pffft_amd import api
if os.environ.get("STOCK_AB_LIB"):
    api.lib_path = lambda: os.path.abspath(os.environ["STOCK_AB_LIB"])

def run(N, tr, dtype, label, ordered=False, direction=pa.FORWARD):
    s = pa.Setup(N, tr, dtype)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    batch = (1 << 29) // (s.vec_scalars * (4 if dtype == np.float32 else 8))
    x = torch.rand(batch, s.vec_scalars, device="cuda", dtype=tdt) * 2 - 1
    y = torch.empty_like(x)
    f = lambda: s.transform_batch(x, y, direction, ordered)
    for _ in range(20): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20 * 1e-3
    byts = 2 * batch * s.vec_scalars * x.element_size()
    print(f"{label:34s} [{pa.kernel_name(s):10s}] {t*1e3:8.3f} ms  frac={byts/t/8e12:.3f}", flush=True)
    del x, y; torch.cuda.empty_cache(); s.close()

CS = tuple(int(v) for v in os.environ.get("STOCK_AB_C", "96,288,384,480,576,768,800,2000,2400,4000,4608,6000,9216").split(","))
RS = tuple(int(v) for v in os.environ.get("STOCK_AB_R", "576,768,960,1152,1536,1600,4000,8000,9216,12000").split(","))
for N in CS:
    run(N, pa.COMPLEX, np.float32, f"cplx f32 N={N}")
    run(N, pa.COMPLEX, np.float32, f"cplx f32 N={N} ordered", ordered=True)
    run(N, pa.COMPLEX, np.float32, f"cplx f32 N={N} bwd", direction=pa.BACKWARD)
for N in RS:
    run(N, pa.REAL, np.float32, f"real f32 N={N}")
    run(N, pa.REAL, np.float32, f"real f32 N={N} ordered", ordered=True)
    run(N, pa.REAL, np.float32, f"real f32 N={N} bwd", direction=pa.BACKWARD)
