"""GPU parity tests (-m gpu), round 3: the whole accepted-size set.

The reference accepts every N = nmin * 2^a * 3^b * 5^c (src/pffft_priv_impl.h:91-114; its own enumeration:
tests/test_fft_factors.c:36-61).  Routing here is six kernel families plus a factor-choice heuristic, so hand-picked
size lists leave room for a size that falls between the families: this file walks EVERY legal size up to 2^18 (and every
7th legal size from there to 2^21), both precisions, real and complex, all four direction x layout combinations on a
ragged batch, against oracle/_ref (the reference's own object code) — and asserts the kernel family each size is
expected to run on, restated here from DESIGN.md §3 independently of the library's planner.
Bars: 1e-5 float / 1e-12 double per transform (tests/conftest.py: tol_for states the one documented exception)."""
import numpy as np
import pytest

from conftest import legal_sizes, relerr, tol_for

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import pffft_amd as pa  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available() or pa.device_count() < 1:
        pytest.fail("GPU tests need a HIP device: the product has no CPU fallback")
    torch.cuda.set_device(0)


def expected_family(N, transform, dt):
    """DESIGN.md §3, restated: which kernel family a size must run on (n = complex points of the core transform)."""
    n = N if transform == pa.COMPLEX else N // 2
    esz = 8 if dt == "f32" else 16
    if n == 16 or (n == 32 and dt == "f32"):
        return "tiny"
    if dt == "f32" and transform == pa.COMPLEX and N == 1024:
        return "c1024_f32"
    if n & (n - 1) == 0:
        return "tiled" if n * esz <= 128 * 1024 else "fourstep"
    # mixed-radix Stockham plans: two exchange images in LDS (n * esz <= 80 000 B); n = 16 * 3^5 (* 2) has no plan in
    # radices 3 .. 24 within four stages: radix 9 (round 4: 3 x 9 x 9 x 16, 6 x 9 x 9 x 16)
    if n * esz <= 80000:
        return "stockham"
    return "fourstep"


def _uniform(shape, seed, tdt):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    x = torch.empty(shape, device="cuda", dtype=tdt)
    x.uniform_(-1.0, 1.0, generator=g)
    return x


def _check_size(ref, N, tr, dt, batch):
    dtype = np.float32 if dt == "f32" else np.float64
    tdt = torch.float32 if dt == "f32" else torch.float64
    s = pa.Setup(N, tr, dtype)
    assert pa.kernel_name(s) == expected_family(N, tr, dt), (dt, tr, N, pa.kernel_name(s))
    rs = ref.setup(N, tr, dtype)
    tol = tol_for(dt, N)
    x = _uniform((batch, s.vec_scalars), 3000 + N % 9973, tdt)
    xh = x.cpu().numpy()
    worst = 0.0
    for d in (pa.FORWARD, pa.BACKWARD):
        for o in (False, True):
            got = s.transform_batch(x, None, d, o).cpu().numpy()
            e = relerr(got, rs.batch(xh, d, o))
            assert e <= tol, (dt, tr, N, d, o, e)
            worst = max(worst, e)
    s.close(); rs.close()
    return worst


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("tr", [pa.COMPLEX, pa.REAL])
def test_every_legal_size_up_to_2p18(ref, dt, tr):
    sizes = legal_sizes(tr, 0, 1 << 18)
    assert len(sizes) > 150 and sizes[0] == (32 if tr == pa.REAL else 16) and sizes[-1] == 1 << 18
    fams = {}
    for N in sizes:
        _check_size(ref, N, tr, dt, batch=3 if N <= 65536 else 2)
        f = expected_family(N, tr, dt)
        fams[f] = fams.get(f, 0) + 1
    # every family is exercised by the walk
    want = {"tiny", "tiled", "stockham", "fourstep"} | ({"c1024_f32"} if (dt == "f32" and tr == pa.COMPLEX) else set())
    assert set(fams) == want, fams


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("tr", [pa.COMPLEX, pa.REAL])
def test_a_stride_of_legal_sizes_up_to_2p21(ref, dt, tr):
    sizes = legal_sizes(tr, (1 << 18) + 1, 1 << 21)[::7]
    assert len(sizes) >= 10
    for N in sizes:
        _check_size(ref, N, tr, dt, batch=2)


# ------------------------------------------------------------------ FIR: one wavefront per block, step = 2048 - taps + 1
@pytest.mark.parametrize("flush", [1, 0])
@pytest.mark.parametrize("taps,L,nsig", [(32, 1 << 22, 1), (33, 3000001, 2), (64, (1 << 21) + 17, 3), (100, 1 << 22, 1),
                                         (255, 3000001, 1), (600, (1 << 22) + 5, 2), (1021, 1 << 22, 1), (1023, 3000001, 1),
                                         (1024, (1 << 22) + 1, 2)])
def test_fastconv_wave_kernel(ref, taps, L, nsig, flush):
    """fastconv_wave_kernel (fft_fir.h, round 3): filters of 32 .. 1024 taps on calls with many blocks - 2048-sample blocks, one
    wavefront each, advancing by the 2048 - taps + 1 (rounded down to 4) samples the filter leaves valid.  Same count as the
    reference's block schedule (src/pffastconv.c:156-166,204-210), values within the reference test's limit
    (tests/test_pffastconv.c:685) over the WHOLE signals, samples beyond the produced ones untouched; the device entry on one
    signal and the batched entry on several."""
    rng = np.random.default_rng(1000 * taps + nsig)
    xs = rng.uniform(-1, 1, (nsig, L)).astype(np.float32)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    fc = pa.FastConv(h, 0, 0)
    xd = torch.from_numpy(xs).cuda()
    yd = torch.full_like(xd, 7.0)
    if nsig == 1:
        y, n = fc.apply(xd[0], bool(flush), out=yd[0])
        got = y.cpu().numpy()[None, :]
    else:
        y, n = fc.apply_batch(xd, bool(flush), out=yd)
        got = y.cpu().numpy()
    for i in range(nsig):
        yw, nw, _ = ref.fastconv(xs[i], h, 0, 0, flush)
        assert n == nw
        assert np.abs(got[i] - yw).max() <= (yw.max() - yw.min()) / 1e5, i
    assert bool((yd[:, n:] == 7.0).all())
    fc.close()


def test_fastconv_batch_more_signals_than_a_grid_dimension_and_stride_checks(ref):
    """the time-domain path took the signal index as blockIdx.y (<= 65535 signals) and accepted an outputStride
    smaller than the samples one signal produces.  70 000 short signals in one call, every one against the reference's
    arithmetic (a direct correlation sum in float64), and the stride checks of include/pffft_hip.h."""
    taps, L, nsig = 5, 64, 70000
    rng = np.random.default_rng(5)
    xs = rng.uniform(-1, 1, (nsig, L)).astype(np.float32)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    fc = pa.FastConv(h, 0, 0)
    xd = torch.from_numpy(xs).cuda()
    y, n = fc.apply_batch(xd, True)
    yw, nw, _ = ref.fastconv(xs[0], h, 0, 0, 1)
    assert n == nw == L - taps + 1
    want = np.zeros((nsig, n))
    for i in range(taps):                                   # y[m] = sum_i h[taps-1-i] x[m+i] (src/pffastconv.c:100-106: the flipped filter)
        want += h[taps - 1 - i].astype(np.float64) * xs[:, i:i + n]
    got = y.cpu().numpy()
    assert np.abs(got[0] - yw).max() <= 1e-5
    assert np.abs(got - want).max() <= 1e-5
    # outputStride smaller than the samples a signal produces: refused (-1), nothing launched
    L2 = pa.lib()
    out = torch.zeros(nsig * 8, device="cuda")
    rc = L2.pffastconv_hip_apply_batch(fc.handle, xd.data_ptr(), L, L, out.data_ptr(), 8, nsig, 1, None)
    assert rc == -1 and b"outputStride" in L2.pffft_hip_last_error()
    assert float(out.abs().max()) == 0.0
    fc.close()


# ------------------------------------------------------------------ sizes with factors 3 / 5 beyond LDS on the tile passes
@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("N", [10240, 14400, 15360, 17280, 36864, 61440, 115200, 368640, 327680, 373248, 1024000])
def test_odd_stage_tile_plans(ref, dt, N):
    """fft_tile.h with an odd first stage (tile lengths 3 / 5 / 9 / 15 x 2^b): two passes where the odd part of n splits over two
    tile lengths (10240 = 64 x 160 ... 368640 = 480 x 768; 327680 = 512 x 640), three where there is no such split and the streaming
    route would need five sweeps (1024000 = 2^13 x 125 = 80 x 160 x 80); 14400 = 120 x 120 and 17280 = 120 x 144 in float have a ragged
    last tile of 8 sequences (a tile is 16), 373248 = 72 x 72 x 72 three ragged passes.  Complex N and real 2N, four direction x layout combinations against oracle/_ref;
    the streaming passes (variant 83) must agree with the tile passes to the same bar - two independent routes, one answer."""
    dtype = np.float32 if dt == "f32" else np.float64
    tdt = torch.float32 if dt == "f32" else torch.float64
    for tr, NN in ((pa.COMPLEX, N), (pa.REAL, 2 * N)):
        _check_size(ref, NN, tr, dt, batch=2)
        s = pa.Setup(NN, tr, dtype)
        x = _uniform((2, s.vec_scalars), 77 + N % 1000, tdt)
        a = s.transform_batch(x, None, pa.FORWARD, True)
        pa.set_variant(83)
        try:
            b = s.transform_batch(x, None, pa.FORWARD, True)
        finally:
            pa.set_variant(0)
        assert relerr(a.cpu().numpy(), b.cpu().numpy()) <= tol_for(dt, NN)
        # in place == out of place, bit for bit, both layouts (the passes go through the work buffer either way)
        for o in (True, False):
            want = s.transform_batch(x, None, pa.FORWARD, o)
            xi = x.clone()
            got = s.transform_batch(xi, xi, pa.FORWARD, o)
            assert torch.equal(got, want), (NN, o)
        s.close()
