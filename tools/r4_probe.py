"""Round-4 A/B on the GPU: (a) n = 8192 float (C3 = real N 16384, and complex) on the multi-wave configurations (variants 100 ..),
(b) FIR 4096 / 2048 taps on the block-kernel organisations (variants 110 ..).  Values are checked first (against variant 0 for the
transforms: different factorisations, 1e-5 relative; against oracle/_ref for the FIR), then fractions of the 8 TB/s roofline."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from dma_ab import timed
from oracle import ref as oref

what = sys.argv[1] if len(sys.argv) > 1 else "mw,fir"
R = oref.get()

if "mw" in what:
    variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,100,101,102,104,107").split(",")]
    for N, tr in ((16384, pa.REAL), (8192, pa.COMPLEX)):
        s = pa.Setup(N, tr)
        B = (1 << 30) // (s.vec_scalars * 4)
        x = torch.rand(B, s.vec_scalars, device="cuda") * 2 - 1
        y = torch.empty_like(x)
        for d in (pa.FORWARD, pa.BACKWARD):
            for o in (False, True):
                row = []
                ref = None
                for var in variants:
                    pa.set_variant(var)
                    s.transform_batch(x, y, d, o)
                    torch.cuda.synchronize()
                    got = torch.cat((y[:3], y[B // 2: B // 2 + 2], y[-3:])).double()
                    if ref is None:
                        ref = got
                        err = 0.0
                    else:
                        err = float(((got - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max())
                    t = min(timed(lambda: s.transform_batch(x, y, d, o), 10) for _ in range(2))
                    row.append(f"v{var}: {2 * x.numel() * 4 / t / 8e12:.3f} ({err:.0e})")
                pa.set_variant(0)
                print(f"N={N:6d} {'cplx' if tr else 'real'} {'fwd' if d == 0 else 'bwd'} {'canonical' if o else 'internal '}: " + "  ".join(row), flush=True)
        del x, y
        s.close()

if "fir" in what:
    variants = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,116,97").split(",")]
    rng = np.random.default_rng(4)
    for taps in (4096, 2048, 1500):
        h = rng.uniform(-1, 1, taps).astype(np.float32)
        x = torch.rand(1 << 26, device="cuda") * 2 - 1
        y = torch.empty_like(x)
        xb = torch.rand(256, 1 << 20, device="cuda") * 2 - 1
        yb = torch.empty_like(xb)
        xs = x[: 1 << 20].cpu().numpy()
        yw, nw, _ = R.fastconv(xs, h, 0, 0, 1)
        for var in variants:
            pa.set_variant(var)
            fc = pa.FastConv(h, 0, 0)
            try:
                ya, n = fc.apply(x[: 1 << 24].contiguous(), True)
                torch.cuda.synchronize()
                err = float(np.abs(ya[: nw].cpu().numpy() - yw).max() / (yw.max() - yw.min()))
                tail = ya[n - 70000: n].cpu().numpy()
                # the last blocks (ragged end) against the time-domain sum on the host
                xe = x[(1 << 24) - 70000 - taps + 1: 1 << 24].cpu().numpy().astype(np.float64)
                want = np.convolve(xe, h.astype(np.float64), mode="valid")
                err2 = float(np.abs(tail - want[-70000:]).max() / (want.max() - want.min())) if n == (1 << 24) - taps + 1 else -1.0
                t = min(timed(lambda: fc.apply(x, True, out=y), 5) for _ in range(2))
                tb = min(timed(lambda: fc.apply_batch(xb, True, out=yb), 5) for _ in range(2))
                print(f"v{var} {taps} taps: 2^26 samples {8 * ((1 << 26) - taps + 1) / t / 8e12:.3f}   256 x 2^20 {8 * 256 * ((1 << 20) - taps + 1) / tb / 8e12:.3f}"
                      f"   err/range {err:.1e} tail {err2:.1e}", flush=True)
            except Exception as e:   # noqa: BLE001
                print(f"v{var} {taps} taps: FAILED {e} | {pa.last_error()}", flush=True)
            fc.close()
        pa.set_variant(0)
        del x, y, xb, yb

if "single" in what:
    # the stated C4 call (2^20 samples, 4096 taps: 255 reference-sized blocks) and two neighbours, back-to-back launches
    rng = np.random.default_rng(4)
    for taps, L in ((4096, 1 << 20), (3000, 1 << 20), (2048, 1 << 19), (1500, 300001), (8192, 1 << 21)):
        h = rng.uniform(-1, 1, taps).astype(np.float32)
        x = torch.rand(L, device="cuda") * 2 - 1
        y = torch.empty_like(x)
        yw, nw, _ = R.fastconv(x.cpu().numpy(), h, 0, 0, 1)
        for var in (0, 115, 114):
            pa.set_variant(var)
            fc = pa.FastConv(h, 0, 0)
            ya, n = fc.apply(x, True, out=y)
            torch.cuda.synchronize()
            err = float(np.abs(ya[:nw].cpu().numpy() - yw).max() / (yw.max() - yw.min())) if n == nw else -1.0
            t = min(timed(lambda: fc.apply(x, True, out=y), 200) for _ in range(3))
            print(f"single call v{var} {taps} taps on {L} samples: {t * 1e6:.2f} us  frac {8 * (L - taps + 1) / t / 8e12:.3f}  err/range {err:.1e} (n {n} / {nw})", flush=True)
            fc.close()
        pa.set_variant(0)
