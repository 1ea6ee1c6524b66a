"""Launches the dominant kernel of every BASELINE config at its stated size a few times (for rocprofv3 passes):
c2 N=1024 cplx f32 2^20, c3 N=16384 real f32 2^16, c5 N=1024 cplx f64 2^20, c4 FIR 2^26 samples / 4096 taps,
beyond-LDS N=2^16 and 2^20 cplx f32 (1 GiB of vectors), N = 600000 = 750 x 800 on the run-time tile passes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
reps = 40   # the first ~15 launches of a kernel climb to the steady rate (c3_ramp.py (earlier-round tool, git history)); pmc_r02.py averages the last ones
def fft(N, tr, dtype, B, ordered=False):
    s = pa.Setup(N, tr, dtype)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    x = torch.empty(B, s.vec_scalars, device="cuda", dtype=tdt).uniform_(-1, 1)
    y = torch.empty_like(x)
    for _ in range(reps + 1):
        s.transform_batch(x, y, pa.FORWARD, ordered)
    torch.cuda.synchronize()
    s.close(); del x, y; torch.cuda.empty_cache()
fft(1024, pa.COMPLEX, np.float32, 1 << 20)
fft(16384, pa.REAL, np.float32, 1 << 16)
fft(1024, pa.COMPLEX, np.float64, 1 << 20)
h = np.random.default_rng(4).uniform(-1, 1, 4096).astype(np.float32)
fc = pa.FastConv(h, 0, 0)
x = torch.empty(1 << 26, device="cuda").uniform_(-1, 1); y = torch.empty_like(x)
for _ in range(reps + 1):
    fc.apply(x, True, out=y)
torch.cuda.synchronize()
xb = torch.empty(256, 1 << 20, device="cuda").uniform_(-1, 1); yb = torch.empty_like(xb)
for _ in range(reps + 1):
    fc.apply_batch(xb, True, out=yb)                          # the same block kernel on 256 signals of 2^20 (tools/pmc_round.py: second half of its dispatches)
torch.cuda.synchronize()
xs = xb[0].contiguous(); ys = torch.empty_like(xs)
for _ in range(4 * reps):
    fc.apply(xs, True, out=ys)                                # the stated C4 call: 255 reference-sized blocks, fused kernel
torch.cuda.synchronize()
fc.close(); del x, y, xb, yb; torch.cuda.empty_cache()
fw = pa.FastConv(np.random.default_rng(5).uniform(-1, 1, 200).astype(np.float32), 0, 0)
x = torch.empty(1 << 26, device="cuda").uniform_(-1, 1); y = torch.empty_like(x)
for _ in range(reps + 1):
    fw.apply(x, True, out=y)                                  # one wavefront per 2048-sample block (round 3)
torch.cuda.synchronize()
fw.close(); del x, y; torch.cuda.empty_cache()
s = pa.Setup(1024, pa.COMPLEX, np.float32)
x = torch.empty(1 << 20, 2048, device="cuda").uniform_(-1, 1); y = torch.empty_like(x)
H = s.transform_batch(x[:1].contiguous(), None, pa.FORWARD, False).reshape(-1).contiguous()
for _ in range(reps + 1):
    s.convolve_batch(x, H, out=y, scaling=1.0 / 1024)         # round 4: forward x H backward in one kernel
torch.cuda.synchronize()
s.close(); del x, y; torch.cuda.empty_cache()
fft(3888, pa.COMPLEX, np.float32, 34521, ordered=False)       # round 4: Stockham plan with a radix-9 stage
fft(1 << 16, pa.COMPLEX, np.float32, 2048, ordered=True)
fft(1 << 16, pa.COMPLEX, np.float32, 2048, ordered=False)     # round 3: the last tile pass stores the internal layout itself
fft(1 << 20, pa.COMPLEX, np.float32, 128, ordered=True)
fft(1 << 18, pa.REAL, np.float32, 1024, ordered=False)        # round 3: pair pass + internal layout as one block-kernel sweep
fft(4000, pa.COMPLEX, np.float32, 1 << 15, ordered=False)     # a mixed-radix Stockham plan (workgroup kernel)
fft(600000, pa.COMPLEX, np.float32, 223, ordered=True)        # round 4: 750 x 800 on the run-time tile passes (fft_tileg.h), 1 GiB of vectors
fft(12000, pa.COMPLEX, np.float32, 11184, ordered=True)       # round 6: the single-image kernel (fft_one.h), 1 GiB of vectors, 32 x 15 x 25 in place
fft(24000, pa.REAL, np.float32, 11184, ordered=False)         # ... real forward into the internal layout: pair pass in place + gather
fft(9216, pa.COMPLEX, np.float64, 7281, ordered=False)        # ... double, four stages, last one into the layout image
fft(1024, pa.COMPLEX, np.float32, 1 << 12)                    # round 5: the short-launch kernel of the headline size (one transform per wavefront)
fft(1024, pa.COMPLEX, np.float32, 1 << 14)
