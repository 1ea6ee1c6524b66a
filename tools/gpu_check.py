"""Quick GPU-side sanity sweep (development tool, not a test): every mode of the HIP path against
oracle/_ref on random data, plus a first timing of the headline kernel.  Run on the GPU box:
    python tools/gpu_check.py [--quick]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from oracle import ref as oref

R = oref.get()
rng = np.random.default_rng(1234)
fails = 0


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def check(tag, got, want, tol):
    global fails
    e = relerr(got, want)
    ok = e <= tol
    if not ok:
        fails += 1
    print(f"{'ok  ' if ok else 'FAIL'} {tag:60s} err={e:.3e}")


def sweep(dtype, transform, sizes, batch=3):
    tol = 1e-5 if dtype == np.float32 else 1e-12
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    for N in sizes:
        try:
            rs = R.setup(N, transform, dtype)
        except ValueError:
            print("skip (reference rejects)", N); continue
        s = pa.Setup(N, transform, dtype)
        nf = s.vec_scalars
        x = rng.uniform(-1, 1, (batch, nf)).astype(dtype)
        xd = torch.from_numpy(x).cuda()
        name = f"{np.dtype(dtype).name} {'real' if transform == pa.REAL else 'cplx'} N={N} [{pa.kernel_name(s)}]"
        try:
            for ordered in (False, True):
                want = rs.batch(x, oref.FORWARD, ordered)
                got = s.transform_batch(xd, None, pa.FORWARD, ordered).cpu().numpy()
                check(f"{name} fwd ordered={int(ordered)}", got, want, tol)
                wb = rs.batch(want, oref.BACKWARD, ordered)
                gb = s.transform_batch(torch.from_numpy(want).cuda(), None, pa.BACKWARD, ordered).cpu().numpy()
                check(f"{name} bwd ordered={int(ordered)}", gb, wb, tol)
            # in place
            buf = xd.clone()
            s.transform_batch(buf, buf, pa.FORWARD, False)
            check(f"{name} fwd in-place", buf.cpu().numpy(), rs.batch(x, oref.FORWARD, False), tol)
            # zreorder + zconvolve
            U = rs.batch(x, oref.FORWARD, False)
            Ud = torch.from_numpy(U).cuda()
            zr = s.zreorder_batch(Ud, None, pa.FORWARD).cpu().numpy()
            want_zr = np.stack([rs.zreorder(U[i], oref.FORWARD) for i in range(batch)])
            check(f"{name} zreorder fwd (exact)", zr, want_zr, 0.0)
            zb = s.zreorder_batch(torch.from_numpy(want_zr).cuda(), None, pa.BACKWARD).cpu().numpy()
            check(f"{name} zreorder bwd (exact)", zb, U, 0.0)
            V = rs.batch(rng.uniform(-1, 1, (batch, nf)).astype(dtype), oref.FORWARD, False)
            acc0 = rng.uniform(-1, 1, (batch, nf)).astype(dtype)
            for accumulate in (True, False):
                want = np.stack([rs.zconvolve(U[i], V[i], acc0[i], 0.37, accumulate) for i in range(batch)])
                ab = torch.from_numpy(acc0.copy()).cuda()
                s.zconvolve_batch(Ud, torch.from_numpy(V).cuda(), ab, 0.37, accumulate)
                check(f"{name} zconvolve acc={int(accumulate)}", ab.cpu().numpy(), want, tol)
        except RuntimeError as e:
            print("ERR ", name, e)
        # legacy host-pointer path, one vector
        if N <= 4096:
            got = s.transform(x[0], pa.FORWARD)
            check(f"{name} legacy host fwd", got, rs.transform_unordered(x[0], oref.FORWARD), tol)
        s.close(); rs.close()


quick = "--quick" in sys.argv
print("device:", torch.cuda.get_device_name(0), "| ref arch:", R.f32.simd_arch(), R.f64.simd_arch())
csz = [16, 32, 48, 64, 80, 96, 128, 160, 240, 256, 480, 512, 1024, 2000, 2592, 4096, 12000, 16384]
rsz = [32, 64, 96, 128, 160, 192, 256, 288, 384, 480, 512, 640, 864, 1024, 2048, 4000, 4096, 8192, 16384, 36864]
if quick:
    csz, rsz = [16, 64, 96, 1024], [32, 64, 96, 1024, 16384]
sweep(np.float32, pa.COMPLEX, csz)
sweep(np.float32, pa.REAL, rsz)
sweep(np.float64, pa.COMPLEX, [16, 64, 96, 1024, 4000, 8192] if not quick else [16, 1024])
sweep(np.float64, pa.REAL, [32, 64, 96, 1024, 4000, 16384] if not quick else [32, 1024])

# generic kernel on N=1024 complex (variant 1)
pa.set_variant(1)
sweep(np.float32, pa.COMPLEX, [1024])
pa.set_variant(0)

# fast convolution vs the reference
for (L, taps, blk, flags) in [(20000, 129, 0, 0), (5000, 37, 512, 0), (50000, 4096, 0, 0), (3000, 31, 0, 1), (3000, 31, 0, 17), (4000, 64, 0, 64)]:
    cpl = 2 if flags & 1 else 1
    x = rng.uniform(-1, 1, L * cpl).astype(np.float32)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    for flush in (1, 0):
        yw, nw, bl = R.fastconv(x, h, blk, flags, flush)
        fc = pa.FastConv(h, blk, flags)
        yg, ng = fc.apply(x, bool(flush))
        tag = f"fastconv L={L} taps={taps} blk={blk}->{fc.block_len} flags={flags} flush={flush} n={ng}/{nw}"
        if ng != nw or fc.block_len != bl:
            fails += 1; print("FAIL", tag, "count/blocklen mismatch")
        else:
            check(tag, yg, yw, 2e-5)
        yd, nd = fc.apply(torch.from_numpy(x).cuda(), bool(flush))
        check(tag + " dev", yd.cpu().numpy(), yw, 2e-5)
        fc.close()

# ---- timing of the headline shape ----
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

batch = 1 << (16 if quick else 19)
s = pa.Setup(1024, pa.COMPLEX)
x = torch.rand(batch, 2048, device="cuda") * 2 - 1
y = torch.empty_like(x)
gb = batch * 16384 / 1e9
for label, fn in [
    ("copy (torch)", lambda: y.copy_(x)),
    ("c1024 fwd unordered", lambda: s.transform_batch(x, y, pa.FORWARD, False)),
    ("c1024 fwd ordered", lambda: s.transform_batch(x, y, pa.FORWARD, True)),
    ("c1024 bwd unordered", lambda: s.transform_batch(x, y, pa.BACKWARD, False)),
    ("c1024 bwd ordered", lambda: s.transform_batch(x, y, pa.BACKWARD, True)),
    ("c1024 fwd unordered in-place", lambda: s.transform_batch(y, y, pa.FORWARD, False)),
]:
    t = timeit(fn)
    print(f"{label:32s} {t*1e3:8.3f} ms  {gb/t:8.1f} GB/s  {batch/t/1e6:8.1f} M transforms/s")
pa.set_variant(1)
t = timeit(lambda: s.transform_batch(x, y, pa.FORWARD, False), 3)
print(f"{'generic fwd unordered':32s} {t*1e3:8.3f} ms  {gb/t:8.1f} GB/s  {batch/t/1e6:8.1f} M transforms/s")
pa.set_variant(0)
print("FAILS:", fails)
sys.exit(1 if fails else 0)
