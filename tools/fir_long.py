"""pffastconv on long signals (throughput regime): Gsamples/s and fraction of the 8 B/sample HBM roofline."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from bench_configs import timed
import os
pa.set_variant(int(os.environ.get('PFV', '0')))
for L, taps in [(1 << 26, int(t)) for t in os.environ.get('TAPS', '4096,1024,256,64').split(',')]:
    x = torch.rand(L, device="cuda") * 2 - 1
    h = np.random.default_rng(0).uniform(-1, 1, taps).astype(np.float32)
    fc = pa.FastConv(h, 0, 0)
    y = torch.empty_like(x)
    t = timed(lambda: fc.apply(x, True, out=y), 5)
    n_out = L - taps + 1
    print(f"signal 2^26, {taps} taps, Nfft={fc.block_len}: {t*1e3:8.3f} ms {n_out/t/1e9:7.2f} Gsamples/s  {8*n_out/t/1e9:7.1f} GB/s = {8*n_out/t/8e12:.3f} of roofline (read x once, write y once)")
    fc.close(); del x, y
