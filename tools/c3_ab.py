"""A/B of float n = 8192 configurations (C3: N = 16384 real; N = 8192 complex): variants 0 (shipped), 54 (4-stage tiled), 75/76 (3-stage tiled)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from bench_configs import run
NN = int(os.environ.get("ABN", "8192"))
for v in [int(a) for a in (sys.argv[1:] or ["0", "54", "75", "76"])]:
    pa.set_variant(v)
    for N, tr in ((NN, pa.COMPLEX), (2 * NN, pa.REAL)):
        s = pa.Setup(N, tr, np.float32)
        x = np.random.default_rng(1).uniform(-1, 1, (3, s.vec_scalars)).astype(np.float32)
        y = s.transform_batch(torch.from_numpy(x).cuda(), None, pa.FORWARD, True)
        yc = y.cpu().numpy().astype(np.float64); yc = yc[:, 0::2] + 1j * yc[:, 1::2]
        if tr == pa.COMPLEX:
            want = np.fft.fft(x[:, 0::2].astype(np.float64) + 1j * x[:, 1::2], axis=1)
        else:
            w = np.fft.rfft(x.astype(np.float64), axis=1); want = w[:, :-1].copy(); want[:, 0] = w[:, 0].real + 1j * w[:, -1].real
        z = s.transform_batch(y, None, pa.BACKWARD, True).cpu().numpy() / N
        yu = s.transform_batch(torch.from_numpy(x).cuda(), None, pa.FORWARD, False)
        zu = s.transform_batch(yu, None, pa.BACKWARD, False).cpu().numpy() / N
        eq = torch.equal(s.zreorder_batch(yu, None, pa.FORWARD), y)
        print(f"variant {v} N={N} tr={tr}: fwd err {np.abs(yc - want).max() / np.abs(want).max():.2e} roundtrip {np.abs(z - x).max():.2e} / {np.abs(zu - x).max():.2e} ordered==zreorder(unordered): {eq}")
        s.close()
    for N, tr, lab in ((NN, pa.COMPLEX, "cplx"), (2 * NN, pa.REAL, "real")):
        for ordered in (False, True):
            for d in (pa.FORWARD, pa.BACKWARD):
                run(N, tr, np.float32, (1 << 31) // (NN * 8), f"v{v} N={N} {lab} f32 {'fwd' if d == pa.FORWARD else 'bwd'} ordered={int(ordered)}", ordered=ordered, direction=d)
