"""Throughput of every BASELINE config + a size sweep (development tool)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3

def run(N, tr, dtype, batch, label, ordered=False, direction=pa.FORWARD, inplace=False):
    s = pa.Setup(N, tr, dtype)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    x = torch.rand(batch, s.vec_scalars, device="cuda", dtype=tdt) * 2 - 1
    y = x if inplace else torch.empty_like(x)
    t = timed(lambda: s.transform_batch(x, y, direction, ordered))
    byts = 2 * batch * s.vec_scalars * x.element_size()
    print(f"{label:46s} [{pa.kernel_name(s):10s}] {t*1e3:9.3f} ms {byts/t/1e9:8.1f} GB/s {batch/t/1e6:9.3f} M/s  frac={byts/t/8e12:.3f}")
    del x, y; torch.cuda.empty_cache(); s.close()

which = (sys.argv[1:] or ["configs", "sweep"]) if __name__ == "__main__" else []
if "configs" in which:
    run(1024, pa.COMPLEX, np.float32, 1 << 20, "C2 N=1024 cplx f32 fwd unordered")
    run(1024, pa.COMPLEX, np.float32, 1 << 20, "C2 N=1024 cplx f32 inv unordered", direction=pa.BACKWARD)
    run(16384, pa.REAL, np.float32, 1 << 16, "C3 N=16384 real f32 fwd unordered")
    run(16384, pa.REAL, np.float32, 1 << 16, "C3 N=16384 real f32 fwd ordered", ordered=True)
    run(1024, pa.COMPLEX, np.float64, 1 << 19, "C5 N=1024 cplx f64 fwd unordered (2^19)")
    run(8192, pa.REAL, np.float32, 1 << 15, "FIR-size N=8192 real f32 fwd")
    run(8192, pa.REAL, np.float32, 1 << 15, "FIR-size N=8192 real f32 bwd", direction=pa.BACKWARD)
    # C4 FIR
    L, taps = 1 << 20, 4096
    x = torch.rand(L, device="cuda") * 2 - 1
    h = np.random.default_rng(0).uniform(-1, 1, taps).astype(np.float32)
    fc = pa.FastConv(h, 0, 0)
    y = torch.empty_like(x)
    t = timed(lambda: fc.apply(x, True, out=y), 20)
    print(f"C4 FIR 2^20 signal, 4096 taps, Nfft={fc.block_len}: {t*1e6:9.1f} us per call, {(L-taps+1)/t/1e9:.3f} Gsamples/s, {8*(L-taps+1)/t/1e9:.1f} GB/s ideal-bytes")
if "sweep" in which:
    for N in (16, 64, 256, 512, 2048, 4096, 8192, 16384):
        b = max(1, (1 << 31) // (N * 8) // 2)
        run(N, pa.COMPLEX, np.float32, b, f"sweep cplx f32 N={N} batch={b}")
    for N in (64, 1024, 4096, 4000, 12000):
        b = max(1, (1 << 31) // (N * 4) // 2)
        run(N, pa.REAL, np.float32, b, f"sweep real f32 N={N} batch={b}")
    for N in (96, 480, 2592):
        b = max(1, (1 << 31) // (N * 8) // 2)
        run(N, pa.COMPLEX, np.float32, b, f"sweep cplx f32 N={N} batch={b}")
