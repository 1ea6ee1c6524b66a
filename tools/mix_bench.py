"""Row f-4 measurements (development tool, GPU box): the mixer kernel alone, and the frequency shift fused into
the N=1024 forward FFT against the two-pass composition and the plain transform."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa
from pffft_amd import pfdsp
from bench_configs import timed

n = 1 << 28                                                    # 2 GiB in, 2 GiB out
x = torch.view_as_complex(torch.rand(n, 2, device="cuda") * 2 - 1)
y = torch.empty_like(x)
t = timed(lambda: pfdsp.shift_device(x, 0.0137, 0.4, out=y))
print(f"mixer kernel, lanes=1, 2^28 samples: {t*1e3:8.3f} ms {16*n/t/1e9:8.1f} GB/s {n/t/1e9:7.2f} Gsamples/s frac={16*n/t/8e12:.3f}")
t = timed(lambda: pfdsp.shift_device(x, 0.0137, 0.4, out=x))
print(f"mixer kernel, in place:              {t*1e3:8.3f} ms {16*n/t/1e9:8.1f} GB/s frac={16*n/t/8e12:.3f}")
t = timed(lambda: y.copy_(x))
print(f"torch copy_ of the same buffers:     {t*1e3:8.3f} ms {16*n/t/1e9:8.1f} GB/s frac={16*n/t/8e12:.3f}")
m = pfdsp.Mixer("recursive_osc", 0.0137, 0.4)
t = timed(lambda: m(x, inplace=True))
print(f"shift_recursive_osc_inp_c (8 lanes), device pointer, incl. host state math + sync: {t*1e3:8.3f} ms {16*n/t/1e9:8.1f} GB/s")
del x, y; torch.cuda.empty_cache()

if "mixonly" in sys.argv:
    sys.exit(0)
N, batch = 1024, 1 << 20
s = pa.Setup(N, pa.COMPLEX, np.float32)
x = torch.rand(batch, 2 * N, device="cuda") * 2 - 1
y = torch.empty_like(x)
byts = 2 * batch * 2 * N * 4
for ordered in (False, True):
    t0 = timed(lambda: s.transform_batch(x, y, pa.FORWARD, ordered))
    t1 = timed(lambda: s.shift_transform_batch(x, 0.0137, 0.4, out=y, ordered=ordered))
    pa.set_variant(60)
    t2 = timed(lambda: s.shift_transform_batch(x, 0.0137, 0.4, out=y, ordered=ordered))
    pa.set_variant(0)
    print(f"N=1024 cplx f32 fwd ordered={int(ordered)}: plain {batch/t0/1e6:7.1f} M/s ({byts/t0/8e12:.3f})  "
          f"shift fused {batch/t1/1e6:7.1f} M/s ({byts/t1/8e12:.3f})  shift two-pass {batch/t2/1e6:7.1f} M/s ({byts/t2/8e12:.3f} of the fused bytes)")
