"""Sizes beyond LDS (four-step path): throughput as a fraction of the single-pass roofline (2 x vector bytes)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from bench_configs import run
for N, tr, dt in ((32768, pa.COMPLEX, np.float32), (65536, pa.COMPLEX, np.float32), (1 << 20, pa.COMPLEX, np.float32), (65536, pa.REAL, np.float32),
                  (1 << 20, pa.REAL, np.float32), (16384, pa.COMPLEX, np.float64), (65536, pa.COMPLEX, np.float64)):
    vb = N * (2 if tr == pa.COMPLEX else 1) * np.dtype(dt).itemsize
    for ordered in (False, True):
        run(N, tr, dt, max(1, (1 << 30) // vb), f"N={N} tr={tr} {np.dtype(dt).name} ordered={int(ordered)}", ordered=ordered)
