import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from dma_ab import timed
from oracle import ref as oref
R = oref.get()
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("PFFASTCONV_HIP"))
rng = np.random.default_rng(4)
for taps in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4096").split(",")]:
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    fc = pa.FastConv(h, 0, 0)
    x = torch.rand(1 << 26, device="cuda") * 2 - 1
    y = torch.empty_like(x)
    t = min(timed(lambda: fc.apply(x, True, out=y), 5) for _ in range(2))
    xs = x[: 1 << 20].cpu().numpy()
    ya, n = fc.apply(x[: 1 << 24].contiguous(), True)
    yw, nw, _ = R.fastconv(xs, h, 0, 0, 1)
    err = float(np.abs(ya[: nw].cpu().numpy()[:100000] - yw[:100000]).max() / (yw.max() - yw.min()))
    xb = torch.rand(256, 1 << 20, device="cuda") * 2 - 1
    yb = torch.empty_like(xb)
    tb = min(timed(lambda: fc.apply_batch(xb, True, out=yb), 5) for _ in range(2))
    print(f"[{tag}] {taps} taps: 2^26 samples {8 * ((1 << 26) - taps + 1) / t / 8e12:.3f}   256 x 2^20 {8 * 256 * ((1 << 20) - taps + 1) / tb / 8e12:.3f}   err/range {err:.1e}", flush=True)
    del x, y, xb, yb; fc.close()
