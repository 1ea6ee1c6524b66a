"""Library load + first launch on a fresh process (compressed offload bundles, pffft_amd/build.py): development tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t0 = time.time()
import pffft_amd as pa
s = pa.Setup(1024, pa.COMPLEX)
x = torch.rand(64, 2048, device="cuda")
y = s.transform_batch(x, None, pa.FORWARD, True); torch.cuda.synchronize()
t1 = time.time()
s2 = pa.Setup(600000, pa.COMPLEX); x2 = torch.rand(2, 1200000, device="cuda"); y2 = s2.transform_batch(x2, None, pa.FORWARD, True); torch.cuda.synchronize()
t2 = time.time()
print(f"library load + first transform {t1 - t0:.3f} s; first beyond-LDS transform {t2 - t1:.3f} s")
