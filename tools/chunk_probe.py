"""Does a beyond-LDS transform run faster when the batch is walked in chunks small enough for the intermediate of its passes to stay in the
256 MiB Infinity Cache?  1 GiB of vectors per timed sweep, as whole-batch launches and as per-chunk launches (pass A, pass B, ... per chunk)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa

def run(N, tr, dtype, chunks_mib):
    s = pa.Setup(N, tr, dtype)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    vb = s.vec_scalars * np.dtype(dtype).itemsize
    batch = max(1, (1 << 30) // vb)
    x = torch.rand((batch, s.vec_scalars), device="cuda", dtype=tdt) * 2 - 1
    y = torch.empty_like(x)
    res = []
    for cm in chunks_mib:
        cb = batch if cm == 0 else max(1, (cm << 20) // vb)
        def f0():
            for b0 in range(0, batch, cb):
                s.transform_batch(x[b0:b0 + cb], y[b0:b0 + cb], pa.FORWARD, True)
        # replayed from a HIP graph: device time only (the per-call host cost of the Python binding would dominate the small chunks)
        st = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            f0(); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=st):
                f0()
        f = g.replay
        for _ in range(3): f()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5): f()
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 5 * 1e-3)
        res.append((cm, 2 * x.numel() * x.element_size() / best / 8e12))
    s.close()
    return res

for spec in sys.argv[1:]:
    N, tr, dt = spec.split(":")
    r = run(int(N), pa.REAL if tr == "r" else pa.COMPLEX, np.float32 if dt == "f32" else np.float64, (0, 256, 128, 64, 32, 16))
    print(f"{spec:>16} fwd ordered: " + "  ".join(f"{'whole' if c == 0 else str(c) + ' MiB'} {v:.3f}" for c, v in r), flush=True)
