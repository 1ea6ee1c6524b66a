"""A/B of the one-thread-per-transform kernel (variant 90) on the minimum sizes + parity against numpy / the shipped kernels."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from bench_configs import run
for dt in (np.float32, np.float64):
    for N, tr in ((16, pa.COMPLEX), (32, pa.COMPLEX), (32, pa.REAL), (64, pa.REAL)):
        if dt == np.float64 and (N, tr) in ((32, pa.COMPLEX), (64, pa.REAL)): continue
        s = pa.Setup(N, tr, dt)
        x = torch.from_numpy(np.random.default_rng(N).uniform(-1, 1, (1003, s.vec_scalars)).astype(dt)).cuda()
        res = {}
        for v in (91, 0):
            pa.set_variant(v)
            fo = s.transform_batch(x, None, pa.FORWARD, True); fu = s.transform_batch(x, None, pa.FORWARD, False)
            bo = s.transform_batch(fo, None, pa.BACKWARD, True); bu = s.transform_batch(fu, None, pa.BACKWARD, False)
            res[v] = (fo, fu, bo, bu)
        pa.set_variant(0)
        errs = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(res[0], res[91])]
        zr = torch.equal(s.zreorder_batch(res[0][1], None, pa.FORWARD), res[0][0])
        print(f"{np.dtype(dt).name} N={N} tr={tr}: rel diff vs shipped fwd ord/unord, bwd ord/unord = {['%.1e' % e for e in errs]} ordered==zreorder(unordered): {zr} roundtrip {float((res[0][2] / N - x).abs().max()):.1e}")
        s.close()
        for v in (0, 92):
            pa.set_variant(v)
            for ordered in (False, True):
                for d in (pa.FORWARD, pa.BACKWARD):
                    run(N, tr, dt, (1 << 30) // (s.vec_scalars * np.dtype(dt).itemsize), f"v{v} {np.dtype(dt).name} N={N} tr={tr} {'fwd' if d == 0 else 'bwd'} ord={int(ordered)}", ordered=ordered, direction=d)
        pa.set_variant(0)
