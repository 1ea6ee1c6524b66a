"""A/B of the mixed-radix Stockham kernel (default) against the in-place radix-2..5 kernel (variant 51) on the
non-power-of-two sizes of the reference's benchmark list (benchmarks/bench_pffft.c:445), and against the
register-tiled kernels (variant 50 routes the power-of-two sizes through the Stockham kernel)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from bench_configs import run
quick = "quick" in sys.argv
variants = [int(v) for v in os.environ.get("PFVS", "0,51").split(",")]
CS = [96, 160, 192, 384, 480, 640, 768, 800, 2400, 9216] if not quick else [96, 480, 2400]
RS = [96, 192, 480, 800, 2400, 9216, 4000, 12000] if not quick else [480, 2400]
for v in variants:
    pa.set_variant(v)
    print(f"--- variant {v}")
    for N in CS: run(N, pa.COMPLEX, np.float32, (1 << 30) // (N * 8), f"cplx f32 N={N}")
    for N in CS[:6:2]: run(N, pa.COMPLEX, np.float32, (1 << 30) // (N * 8), f"cplx f32 N={N} ordered", ordered=True)
    for N in CS[:6:2]: run(N, pa.COMPLEX, np.float32, (1 << 30) // (N * 8), f"cplx f32 N={N} bwd", direction=pa.BACKWARD)
    for N in RS: run(N, pa.REAL, np.float32, (1 << 30) // (N * 4), f"real f32 N={N}")
    for N in RS[:4:2]: run(N, pa.REAL, np.float32, (1 << 30) // (N * 4), f"real f32 N={N} ordered", ordered=True)
    for N in (96, 480, 2400): run(N, pa.COMPLEX, np.float64, (1 << 30) // (N * 16), f"cplx f64 N={N}")
if "pow2" in sys.argv:
    for v in (0, 50):
        pa.set_variant(v)
        print(f"--- variant {v}")
        for N in (64, 256, 1024, 4096): run(N, pa.COMPLEX, np.float32, (1 << 30) // (N * 8), f"cplx f32 N={N}")
        for N in (1024, 4096): run(N, pa.REAL, np.float32, (1 << 30) // (N * 4), f"real f32 N={N} ordered", ordered=True)
