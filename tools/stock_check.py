"""Development check of the mixed-radix Stockham kernel against numpy float64 FFTs (no reference needed).
usage: python tools/stock_check.py [variant]   (variant 50 = also route the power-of-two sizes through it)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
pa.set_variant(variant)
rng = np.random.default_rng(3)
bad = 0
SZ_C = [48, 80, 96, 160, 192, 240, 288, 384, 480, 576, 640, 768, 800, 864, 1200, 2400, 2592, 4000, 4608, 9216]
SZ_R = [96, 160, 192, 288, 384, 480, 576, 640, 800, 960, 1600, 2400, 4000, 4800, 9216, 18432]
if variant == 50:
    SZ_C += [32, 64, 128, 256, 512, 1024, 2048, 4096, 8192]
    SZ_R += [64, 128, 1024, 4096, 16384]
for dtype in (np.float32, np.float64):
    tol = 2e-6 if dtype == np.float32 else 1e-13
    for tr, sizes in ((pa.COMPLEX, SZ_C), (pa.REAL, SZ_R)):
        for N in sizes:
            s = pa.Setup(N, tr, dtype)
            name = pa.kernel_name(s)
            for batch in (1, 7, 131):
                x = rng.uniform(-1, 1, (batch, s.vec_scalars)).astype(dtype)
                xd = torch.from_numpy(x).cuda()
                if tr == pa.COMPLEX:
                    want = np.fft.fft(x[:, 0::2].astype(np.float64) + 1j * x[:, 1::2], axis=1)
                else:
                    full = np.fft.rfft(x.astype(np.float64), axis=1)
                    want = full[:, :-1].copy(); want[:, 0] = full[:, 0].real + 1j * full[:, -1].real
                fo = s.transform_batch(xd, None, pa.FORWARD, True)
                g = fo.cpu().numpy().astype(np.float64)
                e1 = np.abs((g[:, 0::2] + 1j * g[:, 1::2]) - want).max() / np.abs(want).max()
                fu = s.transform_batch(xd, None, pa.FORWARD, False)
                e2 = 0.0 if torch.equal(s.zreorder_batch(fu, None, pa.FORWARD), fo) else 1.0
                bo = s.transform_batch(fo, None, pa.BACKWARD, True).cpu().numpy().astype(np.float64) / N
                e3 = np.abs(bo - x).max()
                bu = s.transform_batch(fu, None, pa.BACKWARD, False).cpu().numpy().astype(np.float64) / N
                e4 = np.abs(bu - x).max()
                ok = e1 <= tol and e2 == 0 and e3 <= 4 * tol and e4 <= 4 * tol
                if not ok:
                    bad += 1
                    print(f"FAIL {dtype.__name__} tr={tr} N={N} batch={batch} [{name}] fwd={e1:.2e} unord={e2} bwd={e3:.2e} bwd_un={e4:.2e}")
                    break
            else:
                print(f"ok   {dtype.__name__} tr={tr} N={N} [{name}] fwd={e1:.1e}")
            s.close()
print("FAILURES", bad)
