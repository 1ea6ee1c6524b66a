import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
N = int(sys.argv[1]); B = int(sys.argv[2])
s = pa.Setup(N, pa.COMPLEX)
x = torch.rand(B, 2 * N, device="cuda") * 2 - 1
y = s.transform_batch(x, None, pa.FORWARD, True).cpu().numpy()
xh = x.cpu().numpy().astype(np.float64)
lens = (pa.lib().pffft_hip_tile_plan)
for b in range(B):
    X = np.fft.fft(xh[b, 0::2] + 1j * xh[b, 1::2])
    got = y[b, 0::2] + 1j * y[b, 1::2]
    e = np.abs(got - X) / np.abs(X).max()
    bad = np.nonzero(e > 1e-4)[0]
    print(b, "max err", e.max(), "bad bins", len(bad), bad[:12], bad[-4:] if len(bad) else "")
