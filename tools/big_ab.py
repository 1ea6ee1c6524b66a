"""Sizes beyond LDS: parity against oracle/_ref and fraction of 8 TB/s (2 x vector bytes) for the tile path (variant 0)
and the round-1 composition (variant 82)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from oracle import ref as oref
from dma_ab import timed, relerr
R = oref.get()
what = sys.argv[1:] or ["parity", "rates"]
if "parity" in what:
    bad = 0
    for dt, dtype, tol in (("f32", np.float32, 1e-5), ("f64", np.float64, 1e-12)):
        for tr in (pa.COMPLEX, pa.REAL):
            logs = (14, 15, 16, 17, 18, 19, 20, 21, 22, 24)
            for lg in logs:
                N = 1 << lg
                s = pa.Setup(N, tr, dtype)
                if pa.kernel_name(s) != "fourstep":
                    s.close(); continue
                rs = R.setup(N, tr, dtype)
                B = 3 if lg <= 20 else 1
                x = np.random.default_rng(lg).uniform(-1, 1, (B, s.vec_scalars)).astype(dtype)
                xd = torch.from_numpy(x).cuda()
                for d in (pa.FORWARD, pa.BACKWARD):
                    for o in (True, False):
                        want = rs.batch(x, d, o)
                        y = s.transform_batch(xd, None, d, o)
                        e = relerr(y.cpu().numpy(), want)
                        z = xd.clone(); s.transform_batch(z, z, d, o)
                        ok = e <= tol and torch.equal(z, y)
                        bad += (not ok)
                        if not ok or (d == 0 and o):
                            print(f"{dt} {'cplx' if tr else 'real'} N=2^{lg} dir={d} ordered={int(o)}: relerr {e:.2e} inplace-equal {torch.equal(z, y)} {'OK' if ok else 'FAIL'}", flush=True)
                s.close(); rs.close()
    print("BIG PARITY", "OK" if bad == 0 else f"FAILED ({bad})", flush=True)
if "rates" in what:
    for dtype in (np.float32, np.float64):
        for tr in (pa.COMPLEX, pa.REAL):
            for lg in (15, 16, 17, 18, 19, 20, 21, 22, 24):
                N = 1 << lg
                s = pa.Setup(N, tr, dtype)
                if pa.kernel_name(s) != "fourstep":
                    s.close(); continue
                tdt = torch.float32 if dtype == np.float32 else torch.float64
                B = max(1, (1 << 30) // (s.vec_scalars * np.dtype(dtype).itemsize))
                x = torch.rand(B, s.vec_scalars, device="cuda", dtype=tdt) * 2 - 1
                y = torch.empty_like(x)
                row = []
                for o in (True, False):
                    for var in (0, 82):
                        pa.set_variant(var)
                        t = timed(lambda: s.transform_batch(x, y, pa.FORWARD, o), 5)
                        row.append(2 * x.numel() * x.element_size() / t / 8e12)
                pa.set_variant(0)
                print(f"{np.dtype(dtype).name} {'cplx' if tr else 'real'} N=2^{lg} batch {B}: canonical tile {row[0]:.3f} r01 {row[1]:.3f} | internal tile {row[2]:.3f} r01 {row[3]:.3f}", flush=True)
                del x, y; s.close(); torch.cuda.empty_cache()
