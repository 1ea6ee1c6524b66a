"""Times every odd-stage tile length of fft_tile.h as the column pass (x a power-of-two row pass of 256) and as the row pass (256
columns first), 256 MiB of vectors per launch, forced through PFFFT_HIP_TILE_MRPLAN; run under rocprofv3 --kernel-trace --stats
(tools/tile_len_times.sh), which prints one line per kernel: the pass costs `tile_cost` (tile_tu.hip) is built from.
    python tools/tile_len_times.py f32|f64"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pffft_amd as pa
dt = np.float64 if sys.argv[1] == 'f64' else np.float32
tdt = torch.float64 if dt == np.float64 else torch.float32
esz = 16 if dt == np.float64 else 8
maxl = {3: 8, 5: 7, 9: 6, 15: 5, 25: 4, 27: 4, 45: 4}
for r0, ml in maxl.items():
    for l in range(3 if (dt == np.float64 and r0 >= 9) else 4, ml + 1):
        L = r0 << l
        N = L * 256
        for plan in (f"{r0},{l},1,8", f"1,8,{r0},{l}"):
            os.environ["PFFFT_HIP_TILE_MRPLAN"] = plan
            s = pa.Setup(N, pa.COMPLEX, dt)
            B = (1 << 28) // (N * esz)
            x = torch.rand(B, 2 * N, device='cuda', dtype=tdt) * 2 - 1; y = torch.empty_like(x)
            for _ in range(10): s.transform_batch(x, y, pa.FORWARD, True)
            torch.cuda.synchronize()
            s.close(); del x, y
