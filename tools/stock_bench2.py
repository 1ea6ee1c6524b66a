"""Throughput of the Stockham kernels on non-power-of-two sizes (2 GiB of vectors per launch)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pffft_amd as pa
from bench_configs import run
pa.set_variant(int(os.environ.get("PFV", "0")))
B = 1 << 31
for N in (96, 160, 192, 384, 480, 640, 768, 800, 960, 2400, 4000, 9216): run(N, pa.COMPLEX, np.float32, B // (N * 8) // 2, f"cplx f32 N={N}")
for N in (96, 480, 800, 2400): run(N, pa.COMPLEX, np.float32, B // (N * 8) // 2, f"cplx f32 N={N} ordered", ordered=True)
for N in (96, 480, 800, 2400): run(N, pa.COMPLEX, np.float32, B // (N * 8) // 2, f"cplx f32 N={N} bwd", direction=pa.BACKWARD)
for N in (96, 192, 480, 800, 1600, 2400, 4000, 9216, 12000): run(N, pa.REAL, np.float32, B // (N * 4) // 2, f"real f32 N={N}")
for N in (96, 480, 2400, 9216): run(N, pa.REAL, np.float32, B // (N * 4) // 2, f"real f32 N={N} ordered", ordered=True)
for N in (480, 2400): run(N, pa.REAL, np.float32, B // (N * 4) // 2, f"real f32 N={N} bwd", direction=pa.BACKWARD)
for N in (96, 480, 800, 2400, 4000): run(N, pa.COMPLEX, np.float64, B // (N * 16) // 2, f"cplx f64 N={N}")
for N in (96, 480, 2400, 8000): run(N, pa.REAL, np.float64, B // (N * 8) // 2, f"real f64 N={N}")
