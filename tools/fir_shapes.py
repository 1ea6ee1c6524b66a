"""FIR throughput against the shape of the call: one long signal of 2^24 .. 2^28 samples, batches of shorter signals (4096 taps)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from dma_ab import timed
taps = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = np.random.default_rng(4).uniform(-1, 1, taps).astype(np.float32)
fc = pa.FastConv(h, 0, 0)
for nsig, L in ((1, 1 << 24), (1, 1 << 26), (1, 1 << 28), (4, 1 << 26), (64, 1 << 22), (256, 1 << 20), (1024, 1 << 18), (16, 1 << 22), (64, 1 << 20)):
    x = torch.rand(nsig, L, device="cuda") * 2 - 1
    y = torch.empty_like(x)
    if nsig == 1:
        f = lambda: fc.apply(x[0], True, out=y[0])
    else:
        f = lambda: fc.apply_batch(x, True, out=y)
    reps = max(5, min(50, int(2e-2 / (nsig * L / 300e9))))
    t = min(timed(f, reps) for _ in range(3))
    print(f"{taps} taps, {nsig} x 2^{int(np.log2(L))}: {t * 1e6:9.1f} us  frac {8 * nsig * (L - taps + 1) / t / 8e12:.3f}  (reps {reps})", flush=True)
    del x, y
