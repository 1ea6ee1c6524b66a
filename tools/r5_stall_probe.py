"""The four first-combination outliers of profiles/r04_scan_beyond_lds_f64.txt (cplx 163840, 360000, 450000; real 230400): the scan's own
sequence (setup, 1 GiB batch, 40 + 10 untimed launches, 20 timed) with EVERY launch timed on its own, for the four combinations in the
scan's order and then the first one again (development tool)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa

cases = [(pa.COMPLEX, 163840), (pa.COMPLEX, 360000), (pa.COMPLEX, 450000), (pa.REAL, 230400), (pa.COMPLEX, 131072)]
for tr, N in cases:
    s = pa.Setup(N, tr, np.float64)
    batch = max(2, (1 << 30) // (s.vec_scalars * 8))
    x = torch.rand(batch, s.vec_scalars, device="cuda", dtype=torch.float64) * 2 - 1
    y = torch.empty_like(x)
    alg = 2 * x.numel() * 8
    for _ in range(40): s.transform_batch(x, y, pa.FORWARD, True)
    out = []
    for d, o in ((pa.FORWARD, True), (pa.FORWARD, False), (pa.BACKWARD, True), (pa.BACKWARD, False), (pa.FORWARD, True)):
        for _ in range(10): s.transform_batch(x, y, d, o)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        ev[0].record()
        for i in range(20):
            s.transform_batch(x, y, d, o)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(20)]
        tot = ev[0].elapsed_time(ev[20])
        out.append(f"{'fwd' if d == pa.FORWARD else 'bwd'}/{'ord' if o else 'uno'} frac {alg * 20 / (tot * 1e-3) / 8e12:.3f} "
                   f"ms min {min(ts):.3f} med {sorted(ts)[10]:.3f} max {max(ts):.3f} first3 {ts[0]:.3f},{ts[1]:.3f},{ts[2]:.3f}")
    print(f"{'cplx' if tr == pa.COMPLEX else 'real'} N={N} [{pa.kernel_name(s)}] plan {pa.tile_plan(N if tr == pa.COMPLEX else N // 2, True)}\n   " + "\n   ".join(out), flush=True)
    del x, y; torch.cuda.empty_cache(); s.close()
