"""A/B of the LDS-DMA staged kernels (fft_dma.h; variants 95 = counted vmcnt, 96 = vmcnt(0), 97 = FIR) against the
register-staged ones (variant 0): parity against oracle/_ref first, then fractions of the 8 TB/s roofline."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from oracle import ref as oref

R = oref.get()
dev = "cuda"


def timed(fn, reps=10, warm=None):
    # the first ~15 launches of a kernel climb to the steady rate (c3_ramp.py (earlier-round tool, git history)): warm up ~30 ms before timing
    for _ in range(warm if warm is not None else max(3, 2 * reps)):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps


def relerr(got, want):
    num = np.abs(got - want).max(axis=1); den = np.abs(want).max(axis=1)
    return float((num / den).max())


def parity():
    bad = 0
    for n in (2048, 4096, 8192):
        for tr in (pa.COMPLEX, pa.REAL):
            N = n if tr == pa.COMPLEX else 2 * n
            s = pa.Setup(N, tr)
            rs = R.setup(N, tr)
            for B in (1, 3, 5, 257, 1031):
                x = torch.rand(B, s.vec_scalars, device=dev) * 2 - 1
                idx = sorted({0, B // 2, B - 1})
                xh = x[idx].cpu().numpy()
                for d in (pa.FORWARD, pa.BACKWARD):
                    for o in (True, False):
                        want = rs.batch(xh, d, o)
                        for var in (95, 96):
                            pa.set_variant(var)
                            y = s.transform_batch(x, None, d, o)
                            e = relerr(y[idx].cpu().numpy(), want)
                            z = x.clone(); s.transform_batch(z, z, d, o)        # in place
                            same = torch.equal(z, y)
                            pa.set_variant(0)
                            y0 = s.transform_batch(x, None, d, o)
                            e0 = float((y - y0).abs().max() / y0.abs().max())
                            ok = e <= 1e-5 and same and e0 <= 1e-5
                            bad += (not ok)
                            if not ok or B == 1031:
                                print(f"n={n} {'cplx' if tr else 'real'} B={B} dir={d} ordered={int(o)} var={var}: relerr {e:.2e} vs-default {e0:.2e} inplace-equal {same} {'OK' if ok else 'FAIL'}", flush=True)
            s.close(); rs.close()
    print("PARITY", "OK" if bad == 0 else f"FAILED ({bad})", flush=True)
    return bad


def rates():
    for n in (8192, 4096, 2048):
        for tr in (pa.REAL, pa.COMPLEX):
            N = n if tr == pa.COMPLEX else 2 * n
            s = pa.Setup(N, tr)
            B = (1 << 32) // (s.vec_scalars * 4)          # 4 GiB of vectors (C3: batch 2^16)
            x = torch.rand(B, s.vec_scalars, device=dev) * 2 - 1
            y = torch.empty_like(x)
            for d in (pa.FORWARD, pa.BACKWARD):
                for o in (False, True):
                    row = []
                    for var in (0, 95, 96):
                        pa.set_variant(var)
                        t = timed(lambda: s.transform_batch(x, y, d, o), 10)
                        row.append(2 * x.numel() * 4 / t / 8e12)
                    pa.set_variant(0)
                    print(f"N={N:6d} {'cplx' if tr else 'real'} {'fwd' if d == 0 else 'bwd'} {'canonical' if o else 'internal '}: "
                          f"register-staged {row[0]:.3f}   dma counted {row[1]:.3f}   dma vmcnt(0) {row[2]:.3f}", flush=True)
            del x, y; s.close()
            torch.cuda.empty_cache()


def fir():
    rng = np.random.default_rng(4)
    bad = 0
    for taps, L, nsig in ((4096, 1 << 20, 8), (1024, 300001, 9), (2048, 1 << 19, 5), (600, 200000, 7)):
        h = rng.uniform(-1, 1, taps).astype(np.float32)
        xs = rng.uniform(-1, 1, (nsig, L)).astype(np.float32)
        fc = pa.FastConv(h, 0, 0)
        xd = torch.from_numpy(xs).cuda()
        for flush in (1, 0):
            pa.set_variant(97)
            y, n = fc.apply_batch(xd, bool(flush))
            got = y.cpu().numpy()
            pa.set_variant(0)
            for i in (0, nsig - 1):
                yw, nw, _ = R.fastconv(xs[i], h, 0, 0, flush)
                lim = (yw.max() - yw.min()) / 1e5
                err = np.abs(got[i] - yw).max() if nw else 0.0
                ok = n == nw and err <= lim
                bad += (not ok)
                print(f"FIR dma taps={taps} L={L} nsig={nsig} flush={flush} sig={i}: n={n}/{nw} err/lim {err/lim if nw else 0:.3f} {'OK' if ok else 'FAIL'}", flush=True)
        fc.close()
    # one long signal, both kernels
    for taps in (4096, 1024, 2048, 600):
        h = rng.uniform(-1, 1, taps).astype(np.float32)
        fc = pa.FastConv(h, 0, 0)
        x = torch.rand(1 << 26, device=dev) * 2 - 1
        y = torch.empty_like(x)
        row = []
        for var in (0, 97):
            pa.set_variant(var)
            t = timed(lambda: fc.apply(x, True, out=y), 5)
            row.append(8 * ((1 << 26) - taps + 1) / t / 8e12)
        pa.set_variant(97)
        ya, _ = fc.apply(x[: 1 << 22].contiguous(), True)
        pa.set_variant(0)
        yb, _ = fc.apply(x[: 1 << 22].contiguous(), True)
        d = float((ya - yb).abs().max() / (yb.max() - yb.min()))
        print(f"FIR 2^26 samples {taps} taps: register-staged {row[0]:.3f}  dma {row[1]:.3f} of the 8 B/sample roofline; dma vs staged diff/range {d:.2e}", flush=True)
        xs = torch.rand(256, 1 << 20, device=dev) * 2 - 1
        ys = torch.empty_like(xs)
        row = []
        for var in (0, 97):
            pa.set_variant(var)
            t = timed(lambda: fc.apply_batch(xs, True, out=ys), 5)
            row.append(8 * 256 * ((1 << 20) - taps + 1) / t / 8e12)
        pa.set_variant(0)
        print(f"FIR 256 x 2^20 samples {taps} taps: register-staged {row[0]:.3f}  dma {row[1]:.3f}", flush=True)
        del x, y, xs, ys; fc.close()
        torch.cuda.empty_cache()
    print("FIR PARITY", "OK" if bad == 0 else f"FAILED ({bad})", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["parity", "rates", "fir"]
    if "parity" in what: parity()
    if "rates" in what: rates()
    if "fir" in what: fir()
