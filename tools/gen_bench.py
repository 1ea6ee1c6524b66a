import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pffft_amd as pa
from bench_configs import run
pa.set_variant(int(os.environ.get("PFV", "0")))
for N in (96, 480, 2592): run(N, pa.COMPLEX, np.float32, (1 << 31) // (N * 8) // 2, f"cplx N={N}")
for N in (4000, 12000): run(N, pa.REAL, np.float32, (1 << 31) // (N * 4) // 2, f"real N={N}")
