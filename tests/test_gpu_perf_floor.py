"""Performance floors of the configs the bench reports (-m gpu, < 20 s): fractions of the 8 TB/s roofline on the algorithmic bytes of
SURVEY.md 8(d), best of three timed runs per config, floors ~10 % under the driver-run figures of BENCH_r05 / this round's runs - so that a
regression like round 4's (a static start that cost the in-order Stockham plans 10-12 %, found by bisecting a day's commits) fails the
driver's pytest instead of waiting for someone to read BENCH_r*.json.  Not a benchmark: bench.py is."""
import numpy as np
import pytest

import pffft_amd as pa

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

PEAK = 8e12


def _best(f, reps=5, rounds=5):
    # (round 6: two warm-up launches and three rounds let a cold box fail the convolution floor once in three full-suite runs - the clocks take
    #  ~30 ms of work to come up, bench.py's first_launches_ms shows the ramp; ten launches and five rounds now)
    for _ in range(10): f()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): f()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps * 1e-3)
    return best


def _frac_transform(N, tr, dtype, batch, ordered=False, direction=pa.FORWARD):
    s = pa.Setup(N, tr, dtype)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    x = torch.rand((batch, s.vec_scalars), device="cuda", dtype=tdt) * 2 - 1
    y = torch.empty_like(x)
    t = _best(lambda: s.transform_batch(x, y, direction, ordered))
    s.close()
    return 2 * x.numel() * x.element_size() / t / PEAK


@pytest.mark.parametrize("name,args,floor", [
    ("C2 N=1024 complex float forward, batch 2^18", (1024, pa.COMPLEX, np.float32, 1 << 18), 0.74),
    ("C3 N=16384 real float forward, batch 2^14", (16384, pa.REAL, np.float32, 1 << 14), 0.64),
    ("C5 N=1024 complex double forward, batch 2^18", (1024, pa.COMPLEX, np.float64, 1 << 18), 0.72),
    ("N=2^16 complex float forward (two tile passes), 1 GiB", (1 << 16, pa.COMPLEX, np.float32, 1 << 10), 0.32),
])
def test_transform_floor(name, args, floor):
    f = _frac_transform(*args)
    assert f >= floor, f"{name}: {f:.3f} of the roofline, floor {floor}"


def test_fused_convolution_floor():
    s = pa.Setup(1024, pa.COMPLEX)
    x = torch.rand((1 << 18, 2048), device="cuda") * 2 - 1
    H = s.transform_batch(x[:1].contiguous(), None, pa.FORWARD, False)[0].contiguous()
    y = torch.empty_like(x)
    t = _best(lambda: s.convolve_batch(x, H, y, 1.0 / 1024))
    s.close()
    f = 2 * x.numel() * 4 / t / PEAK
    assert f >= 0.64, f"pffft_hip_convolve_batch N=1024: {f:.3f} of the roofline, floor 0.64"


def test_c4_fir_batch_floor():
    """BASELINE configs[3] in the throughput regime: 64 signals of 2^20 samples, 4096 taps, 8 B per output sample (round 5: 0.35-0.38 on the
    split kernel; round 6: 0.40-0.42 on fft_fir32.h)."""
    rng = np.random.default_rng(4)
    taps, nsig, L = 4096, 64, 1 << 20
    fc = pa.FastConv(rng.uniform(-1, 1, taps).astype(np.float32), 0, 0)
    x = torch.rand((nsig, L), device="cuda") * 2 - 1
    y = torch.empty_like(x)
    t = _best(lambda: fc.apply_batch(x, True, out=y))
    fc.close()
    f = 8.0 * nsig * (L - taps + 1) / t / PEAK
    assert f >= 0.33, f"C4 batch: {f:.3f} of the roofline, floor 0.33"
