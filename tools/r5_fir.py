"""FIR (C4 shapes) with the XCD-contiguous block ranges on / off (PFFASTCONV_HIP_XCD, read once per process: run twice) (development tool)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from r4_graph import per_call
tag = f"XCD={os.environ.get('PFFASTCONV_HIP_XCD', '1')}"
rng = np.random.default_rng(4)
for taps in (4096, 2048):
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    fc = pa.FastConv(h, 0, 0)
    res = []
    for nsig, L in ((1, 1 << 20), (1, 1 << 26), (256, 1 << 20)):
        x = torch.rand(nsig, L, device="cuda") * 2 - 1
        y = torch.empty_like(x)
        f = (lambda: fc.apply(x[0], True, out=y[0])) if nsig == 1 else (lambda: fc.apply_batch(x, True, out=y))
        for _ in range(3): f()
        reps = 300 if L == 1 << 20 and nsig == 1 else 30
        t = min(per_call(f, reps) for _ in range(3))
        res.append(f"{nsig} x 2^{int(np.log2(L))}: {t:.1f} us frac {8 * nsig * (L - taps + 1) / t / 8e6:.3f}")
        del x, y
    print(f"[{tag}] {taps} taps: " + "   ".join(res), flush=True)
    fc.close()
