"""Per-call cost of the legacy single-vector entries on HOST pointers (the reference API as a CPU caller uses it)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import pffft_amd as pa
import ctypes as C
L = pa.lib()
for N, tr, dt in ((64, pa.REAL, np.float32), (1024, pa.COMPLEX, np.float32), (1024, pa.COMPLEX, np.float64), (16384, pa.REAL, np.float32),
                  (65536, pa.COMPLEX, np.float32)):
    s = pa.Setup(N, tr, dt)
    x = pa.api._aligned_empty(s.vec_scalars, dt); x[:] = np.random.default_rng(0).uniform(-1, 1, s.vec_scalars)
    y = pa.api._aligned_empty(s.vec_scalars, dt)
    fn = getattr(L, ("pffftd" if dt == np.float64 else "pffft") + "_transform")
    for _ in range(20): fn(s.handle, x.ctypes.data, y.ctypes.data, None, 0)
    reps = 2000
    t0 = time.perf_counter()
    for _ in range(reps): fn(s.handle, x.ctypes.data, y.ctypes.data, None, 0)
    t = (time.perf_counter() - t0) / reps
    print(f"N={N} tr={tr} {np.dtype(dt).name}: {t*1e6:7.1f} us per pffft_transform call on host pointers ({1/t/1e6:.4f} M transforms/s)")
    s.close()
