"""A/B of the N=1024 complex double configurations (variants 0, 70, 71, 72) + parity against numpy."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from bench_configs import run
NN = int(os.environ.get("ABN", "1024"))
for v in [int(a) for a in (sys.argv[1:] or ["0", "70", "71", "72"])]:
    pa.set_variant(v)
    s = pa.Setup(NN, pa.COMPLEX, np.float64)
    x = np.random.default_rng(1).uniform(-1, 1, (5, 2 * NN))
    for ordered in (True, False):
        y = s.transform_batch(torch.from_numpy(x).cuda(), None, pa.FORWARD, ordered)
        if ordered:
            yc = y.cpu().numpy(); yc = yc[:, 0::2] + 1j * yc[:, 1::2]
            want = np.fft.fft(x[:, 0::2] + 1j * x[:, 1::2], axis=1)
            print("variant", v, "fwd err", np.abs(yc - want).max() / np.abs(want).max())
        z = s.transform_batch(y, None, pa.BACKWARD, ordered).cpu().numpy() / NN
        print("variant", v, "ordered", ordered, "roundtrip err", np.abs(z - x).max())
    s.close()
    for real in (False, True):
      for ordered in (False, True):
        for d in (pa.FORWARD, pa.BACKWARD):
            N = 2 * NN if real else NN
            run(N, pa.REAL if real else pa.COMPLEX, np.float64, (1 << 29) // NN, f"v{v} N={N} {'real' if real else 'cplx'} f64 {'fwd' if d == pa.FORWARD else 'bwd'} ordered={int(ordered)}", ordered=ordered, direction=d)
