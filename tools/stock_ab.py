import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pffft_amd as pa
from bench_configs import run
for v in [int(x) for x in sys.argv[1].split(",")]:
    pa.set_variant(v)
    for spec in sys.argv[2:]:
        f = spec.split(":")
        N, tr, o = int(f[0]), f[1], f[2]
        dbl = len(f) > 3 and f[3] == "d"
        trn = pa.COMPLEX if tr == "c" else pa.REAL
        esz = (8 if tr == "c" else 4) * (2 if dbl else 1)
        batch = (1 << int(f[4])) if len(f) > 4 else (1 << 30) // (N * esz)
        run(N, trn, np.float64 if dbl else np.float32, batch, f"v{v} {tr} N={N} ord={o} {'f64' if dbl else 'f32'} b={batch}", ordered=(o == "1"))
