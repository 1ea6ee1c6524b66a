"""Real forward transforms beyond LDS: the pair pass inside the last tile pass (fft_tile.h RMODE 3, the default where the plan allows) against the
complex core + pair sweep (variant 121), same process, alternating; fractions of 8 TB/s at 1 GiB per launch, values of the two routes compared."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa

def t(f):
    for _ in range(4): f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): f()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10 * 1e-3)
    return best

for spec in sys.argv[1:]:
    N, dt = spec.split(":")
    N = int(N); dtype = np.float32 if dt == "f32" else np.float64
    s = pa.Setup(N, pa.REAL, dtype)
    if "real-rows" not in pa.describe(s):
        print(f"{spec}: no fused row pass for this size"); s.close(); continue
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    batch = max(1, (1 << 30) // (N * np.dtype(dtype).itemsize))
    x = torch.rand((batch, N), device="cuda", dtype=tdt) * 2 - 1
    y = torch.empty_like(x); y2 = torch.empty_like(x)
    row = []
    for o in (True,):
        f = lambda: s.transform_batch(x, y, pa.FORWARD, o)
        tn = t(f)
        pa.set_variant(121)
        g = lambda: s.transform_batch(x, y2, pa.FORWARD, o)
        to = t(g)
        pa.set_variant(0)
        err = float((y - y2).abs().max() / y2.abs().max())
        bytes_ = 2 * x.numel() * x.element_size()
        row.append(f"{'ordered' if o else 'unordered'}: three sweeps {bytes_ / to / 8e12:.3f} -> fused {bytes_ / tn / 8e12:.3f} (rel diff {err:.1e})")
    print(f"{spec:>12}: " + "   ".join(row), flush=True)
    s.close()
