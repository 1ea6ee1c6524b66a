import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from bench_configs import run
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
sizes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [512, 1024, 2048, 4096, 8192, 16384]
dts = (np.float32, np.float64) if len(sys.argv) <= 3 else (np.float32,)
for dt in dts:
    for n in sizes:
        esz = 8 if dt == np.float32 else 16
        if n * esz > 128 * 1024: continue
        b = max(1, (1 << 31) // (n * esz))
        for (tr, N, lab) in ((pa.COMPLEX, n, "cplx"), (pa.REAL, 2 * n, "real")):
            for ordered in (True, False):
                for d in (pa.FORWARD, pa.BACKWARD):
                    for v in variants:
                        pa.set_variant(v)
                        try:
                            run(N, tr, dt, b, f"v{v} {np.dtype(dt).name} {lab} n={n} {'ord' if ordered else 'int'} {'fwd' if d == 0 else 'bwd'}", ordered, d)
                        except Exception as e:
                            print("ERR", n, lab, e)
pa.set_variant(0)
