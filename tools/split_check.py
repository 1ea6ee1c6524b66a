import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from dma_ab import timed, relerr
from oracle import ref as oref
R = oref.get()
N = 16384
s = pa.Setup(N, pa.REAL)
rs = R.setup(N, pa.REAL)
bad = 0
for var in (89, 90):
    for B in (1, 3, 257, 1031):
        x = torch.rand(B, N, device="cuda") * 2 - 1
        idx = sorted({0, B // 2, B - 1})
        xh = x[idx].cpu().numpy()
        for o in (False, True):
            want = rs.batch(xh, 0, o)
            pa.set_variant(var)
            y = s.transform_batch(x, None, pa.FORWARD, o)
            z = x.clone(); s.transform_batch(z, z, pa.FORWARD, o)
            pa.set_variant(0)
            e = relerr(y[idx].cpu().numpy(), want)
            ok = e <= 1e-5 and torch.equal(z, y)
            bad += (not ok)
            print(f"split var={var} B={B} ordered={int(o)}: relerr {e:.2e} inplace-equal {torch.equal(z, y)} {'OK' if ok else 'FAIL'}", flush=True)
print("SPLIT PARITY", "OK" if bad == 0 else f"FAILED ({bad})", flush=True)
B = 1 << 16
x = torch.rand(B, N, device="cuda") * 2 - 1
y = torch.empty_like(x)
for o in (False, True):
    row = []
    for var in (0, 89, 90):
        pa.set_variant(var)
        t = min(timed(lambda: s.transform_batch(x, y, pa.FORWARD, o), 10) for _ in range(3))
        row.append(f"v{var}: {2 * x.numel() * 4 / t / 8e12:.3f}")
    pa.set_variant(0)
    print(f"C3 fwd {'canonical' if o else 'internal '}: " + "  ".join(row), flush=True)
