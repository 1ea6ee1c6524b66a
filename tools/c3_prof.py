import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from bench_configs import run
for N, tr, lab in ((16384, pa.REAL, "real"), (8192, pa.COMPLEX, "cplx"), (4096, pa.COMPLEX, "cplx"), (2048, pa.COMPLEX, "cplx")):
    for ordered in (False, True):
        for d in (pa.FORWARD, pa.BACKWARD):
            run(N, tr, np.float32, (1 << 30) // (N * (8 if tr == pa.COMPLEX else 4)), f"N={N} {lab} {'fwd' if d == 0 else 'bwd'} ord={int(ordered)}", ordered=ordered, direction=d)
