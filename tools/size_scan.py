"""Scan of the legal sizes for slow outliers AND wrong values (development tool):

    python tools/size_scan.py LO HI [f32|f64] [stride] [steady]
    python tools/size_scan.py sizes N1,N2,... [f32|f64] [steady]         (explicit list; sizes a transform type rejects are skipped)

every `stride`-th legal size N = nmin 2^a 3^b 5^c in [LO, HI] (the set tests/test_fft_factors.c:36-61 enumerates), complex and
real, the four direction x layout combinations, 256 MiB per launch, 10 untimed + 10 timed launches -> fraction of 8 TB/s on
2 x vector bytes (short runs: below steady state; the point is outliers).  With a trailing `steady`: 1 GiB per launch, 40 untimed
launches per size, then 10 untimed + 20 timed per combination - the convention of bench.py's sizes table.  While it times it CHECKS: the canonical forward
spectrum of two vectors of the batch against a float64 numpy FFT (1e-5 float / 1e-12 double, the parity bar),
zreorder(unordered) == ordered bit for bit, and the backward transforms through the round trip.  A failed check prints
"BAD" and the exit status is 1.  (No oracle/ import here: tools are not test infrastructure.)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pffft_amd as pa


def legal_sizes(transform, lo, hi):
    nmin = 32 if transform == pa.REAL else 16
    out, a = [], 1
    while nmin * a <= hi:
        b = a
        while nmin * b <= hi:
            c = b
            while nmin * c <= hi:
                if nmin * c >= lo: out.append(nmin * c)
                c *= 5
            b *= 3
        a *= 2
    return sorted(out)


def canonical_f64(x, N, tr):
    """float64 canonical spectrum in pffft's format (include/pffft/pffft.h:144-155)."""
    if tr == pa.COMPLEX:
        X = np.fft.fft(x[0::2].astype(np.float64) + 1j * x[1::2].astype(np.float64))
        out = np.empty(2 * N); out[0::2] = X.real; out[1::2] = X.imag
        return out
    X = np.fft.rfft(x.astype(np.float64))
    out = np.empty(N); out[0] = X[0].real; out[1] = X[N // 2].real
    out[2::2] = X[1:N // 2].real; out[3::2] = X[1:N // 2].imag
    return out


def main():
    explicit = None
    steady = sys.argv[-1] == "steady"
    if steady: sys.argv.pop()
    if sys.argv[1] == "sizes":
        explicit = [int(v) for v in sys.argv[2].split(",")]
        lo, hi = min(explicit), max(explicit)
    else:
        lo, hi = int(sys.argv[1]), int(sys.argv[2])
    dt = np.float64 if len(sys.argv) > 3 and sys.argv[3] == "f64" else np.float32
    stride = int(sys.argv[4]) if len(sys.argv) > 4 and not explicit else 1
    tdt = torch.float64 if dt == np.float64 else torch.float32
    tol = 1e-12 if dt == np.float64 else 1e-5
    bad = 0
    print(f"# tools/size_scan.py {lo} {hi} {np.dtype(dt).name} stride {stride}{' steady (1 GiB per launch, 40 + 10 untimed, 20 timed)' if steady else ''}: fraction of 8 TB/s, "
          "fwd ordered / fwd unordered / bwd ordered / bwd unordered; err = max rel error of the checked vectors vs float64 numpy")
    for tr, name in ((pa.COMPLEX, "cplx"), (pa.REAL, "real")):
        for N in ([v for v in explicit if v in set(legal_sizes(tr, lo, hi))] if explicit else legal_sizes(tr, lo, hi)[::stride]):
            s = pa.Setup(N, tr, dt)
            isz = np.dtype(dt).itemsize
            batch = max(2, (1 << (30 if steady else 28)) // (s.vec_scalars * isz))
            x = torch.rand(batch, s.vec_scalars, device="cuda", dtype=tdt) * 2 - 1
            y = torch.empty_like(x)
            res = []
            hic = 0
            nt = 20 if steady else 10
            if steady:
                for _ in range(40): s.transform_batch(x, y, pa.FORWARD, True)
            for d in (pa.FORWARD, pa.BACKWARD):
                for o in (True, False):
                    f = lambda: s.transform_batch(x, y, d, o)
                    for _ in range(10): f()
                    torch.cuda.synchronize()
                    # every launch between its own pair of events (round 5): the fraction is that of the whole timed region as before; a
                    # launch that took more than 1.6 x the median of its region is a HICCUP of the box or the runtime, not the kernel's
                    # rate - counted per size (hic) and, where there is one, the region is timed once more so that the table shows the
                    # kernel (r04's f64 scan carried four regions at 0.10-0.14 beside 0.24-0.35; tools/r5_stall_probe.py could not
                    # reproduce them with the scan's own sequence)
                    for attempt in range(2):
                        ev = [torch.cuda.Event(enable_timing=True) for _ in range(nt + 1)]
                        ev[0].record()
                        for i in range(nt):
                            f(); ev[i + 1].record()
                        torch.cuda.synchronize()
                        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(nt))
                        slow = sum(1 for v in ts if v > 1.6 * ts[nt // 2])
                        hic += slow
                        if not slow: break
                    res.append(2 * x.numel() * isz / (ev[0].elapsed_time(ev[nt]) / nt * 1e-3) / 8e12)
            # ---- values: first and last vector of the batch
            idx = torch.tensor([0, batch - 1], device="cuda")
            xs = x[idx].contiguous()
            fo = s.transform_batch(xs, None, pa.FORWARD, True)
            fu = s.transform_batch(xs, None, pa.FORWARD, False)
            xh = xs.cpu().numpy()
            err = 0.0
            for i in range(2):
                want = canonical_f64(xh[i], N, tr)
                err = max(err, float(np.abs(fo[i].cpu().numpy() - want).max() / np.abs(want).max()))
            ok = err <= tol
            ok &= bool(torch.equal(s.zreorder_batch(fu, None, pa.FORWARD), fo))      # unordered = permuted ordered, bit for bit
            for o, spec in ((True, fo), (False, fu)):                                # round trips: backward of both layouts
                back = s.transform_batch(spec, None, pa.BACKWARD, o)
                e2 = float((back / N - xs).abs().max())
                err = max(err, e2 / 4)       # round trip of two transforms on |x| <= 1: reported on the scale of one
                ok &= e2 <= 8 * tol
            bad += 0 if ok else 1
            print(f"{name} N={N:8d} [{pa.kernel_name(s):9s}] {res[0]:.3f} {res[1]:.3f} {res[2]:.3f} {res[3]:.3f}  err {err:.1e} "
                  f"{'ok' if ok else 'BAD'}{' hiccups ' + str(hic) if hic else ''}", flush=True)
            del x, y; torch.cuda.empty_cache(); s.close()
    print(f"# {bad} sizes failed the value checks")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
