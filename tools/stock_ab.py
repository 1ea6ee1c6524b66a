import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pffft_amd as pa
from bench_configs import run
for v in [int(x) for x in sys.argv[1].split(",")]:
    pa.set_variant(v)
    for spec in sys.argv[2:]:
        N, tr, o = spec.split(":")
        N = int(N); trn = pa.COMPLEX if tr == "c" else pa.REAL
        esz = 8 if tr == "c" else 4
        run(N, trn, np.float32, (1 << 30) // (N * esz), f"v{v} {tr} N={N} ord={o}", ordered=(o == "1"))
