import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
v = int(sys.argv[1]); N = int(sys.argv[2]); batch = int(sys.argv[3])
pa.set_variant(v)
s = pa.Setup(N, pa.COMPLEX, np.float32)
x = torch.rand(batch, 2 * N, device="cuda") * 2 - 1
y = s.transform_batch(x, None, pa.FORWARD, True)
torch.cuda.synchronize()
want = torch.fft.fft(torch.view_as_complex(x.view(batch, N, 2)), dim=1)
got = torch.view_as_complex(y.view(batch, N, 2))
print("variant", v, "N", N, "batch", batch, pa.kernel_name(s), "err", float((got - want).abs().max() / want.abs().max()))
