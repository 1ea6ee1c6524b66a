"""Round-4 rows: the split FIR block kernel (fft_split.h), the 512-thread few-block configurations, the fused spectral
convolution entry, long-batch parity of the in-order families.  All through the C ABI against oracle/_ref."""
import numpy as np
import pytest

import pffft_amd as pa

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    from oracle import ref as oref
    if not oref.available():
        pytest.skip("oracle/_ref not built")
    return oref.get()


# ------------------------------------------------------------------ split FIR block kernel (fft_split.h)
@pytest.mark.parametrize("taps,L,nsig", [(4096, (1 << 22) + 4321, 1), (851, (1 << 22) + 3, 2), (1025, (1 << 22) + 17, 2), (2048, 3 * (1 << 20) + 3, 3),
                                        (3001, 1 << 22, 2), (8192, (1 << 22) + 5, 1), (5000, 2500001, 2)])
@pytest.mark.parametrize("flush", [1, 0])
def test_fastconv_split_kernel(ref, taps, L, nsig, flush):
    """fastconv_split_kernel - default for 16384-sample internal blocks (filters beyond 850 taps on calls with many blocks) and
    for reference-sized blocks of that length (4097 .. 8192 taps): cross-wave radix 8 + wave-local 1024-point transforms, mirror
    exchange by pairwise flags, LDS-DMA pieces spread over the wave-local phases.  Variant 116 = plain barriers / pieces at once,
    97 = the lock-step kernel it replaced.  Count = the reference's block schedule, values within the reference test's limit over
    the WHOLE signals, samples beyond the produced ones untouched, ragged last block (signal lengths that are not multiples of 4)."""
    rng = np.random.default_rng(taps + nsig + flush)
    xs = rng.uniform(-1, 1, (nsig, L)).astype(np.float32)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    fc = pa.FastConv(h, 0, 0)
    xd = torch.from_numpy(xs).cuda()
    want = [ref.fastconv(xs[i], h, 0, 0, flush) for i in range(nsig)]
    try:
        for var in ((0, 116, 97) if pa.has_variants() else (0,)):      # (116, 97: development-build kernels, pf_route.h AbValue)
            pa.set_variant(var)
            yd = torch.full_like(xd, 7.0)
            y, n = fc.apply_batch(xd, bool(flush), out=yd)
            got = y.cpu().numpy()
            for i in range(nsig):
                yw, nw, _ = want[i]
                assert n == nw, (var, i)
                assert np.abs(got[i] - yw).max() <= (yw.max() - yw.min()) / 1e5, (var, i)
            assert bool((yd[:, n:] == 7.0).all()), var
    finally:
        pa.set_variant(0)
    fc.close()


@pytest.mark.parametrize("taps,L", [(4096, 1 << 20), (2048, 1 << 19), (1500, 300001), (3000, 700003), (2049, 8192 * 3 + 5), (1025, 40001),
                                    (4096, 5 * 4097 * 300 + 77)])
def test_fastconv_few_blocks_512_thread_configurations(ref, taps, L):
    """Calls with few blocks (the stated C4 call: 2^20 samples, 4096 taps) run one workgroup per reference-sized block; round 4:
    512 / 256 threads per 8192- / 4096-sample block (FirCfg::C4096m / C2048m, eight points per thread), the first block's samples
    requested before the tables (variant 115), and - the default since - cross-wave radix 8 / 4 + wave-local 512-point transforms
    (fastconv_split1_kernel).  Variant 114 = the 16-points-per-thread configurations."""
    rng = np.random.default_rng(taps)
    x = rng.uniform(-1, 1, L).astype(np.float32)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    fc = pa.FastConv(h, 0, 0)
    xd = torch.from_numpy(x).cuda()
    try:
        for flush in (1, 0):
            yw, nw, _ = ref.fastconv(x, h, 0, 0, flush)
            for var in ((0, 115, 114) if pa.has_variants() else (0, 115)):   # (115: the lock-step kernel, the second route; 114: development build)
                pa.set_variant(var)
                yd = torch.full_like(xd, 7.0)
                y, n = fc.apply(xd, bool(flush), out=yd)
                assert n == nw, (var, flush)
                assert np.abs(y.cpu().numpy() - yw).max() <= (yw.max() - yw.min()) / 1e5, (var, flush)
                assert bool((yd[n:] == 7.0).all())
    finally:
        pa.set_variant(0)
    fc.close()


# ------------------------------------------------------------------ pffft_hip_convolve_batch (fft_conv.h + composition)
def _ref_convolve(rs, x2d, H2d, scaling, acc0=None):
    """The reference's sequence, vector by vector: transform FORWARD, zconvolve_no_accu, transform BACKWARD (+ acc0)."""
    from oracle.ref import FORWARD, BACKWARD
    out = np.empty_like(x2d)
    for i in range(x2d.shape[0]):
        X = rs.transform_unordered(x2d[i], FORWARD)
        Hh = H2d[i if H2d.shape[0] > 1 else 0]
        Y = rs.zconvolve(X, Hh, np.zeros_like(X), scaling, accumulate=False)
        out[i] = rs.transform_unordered(Y, BACKWARD)
    return out if acc0 is None else out + acc0


@pytest.mark.parametrize("dt,tol", [(np.float32, 1e-5), (np.float64, 1e-12)])
@pytest.mark.parametrize("tr,N", [(pa.COMPLEX, 16), (pa.COMPLEX, 64), (pa.COMPLEX, 256), (pa.COMPLEX, 1024), (pa.COMPLEX, 4096),
                                  (pa.COMPLEX, 8192), (pa.REAL, 32), (pa.REAL, 64), (pa.REAL, 128), (pa.REAL, 512), (pa.REAL, 2048),
                                  (pa.REAL, 8192), (pa.REAL, 16384), (pa.COMPLEX, 96), (pa.REAL, 1920), (pa.COMPLEX, 32768),
                                  (pa.REAL, 1 << 17)])
def test_convolve_batch_against_the_reference_sequence(ref, dt, tol, tr, N):
    """out = backward(forward(in) . H) scaling against pffft_transform / pffft_zconvolve_no_accu / pffft_transform of the reference,
    per vector: fused kernel (power-of-two sizes, broadcast H), the composition it falls back to (variant 120, sizes with factors
    3 / 5, beyond LDS, per-vector H), accumulate, in place.  Ragged batch (more vectors than one grid of workgroups takes at once
    is covered by the C2 / C5 shapes below)."""
    from oracle.ref import FORWARD
    rs = ref.setup(N, tr, dt)
    s = pa.Setup(N, tr, dt)
    rng = np.random.default_rng(N + tr)
    B = 5
    if dt == np.float64 and N & (N - 1):
        tol = 2e-7     # the reference's double build keeps float-suffixed radix-3 / 5 constants: it is the inexact side
    x = rng.uniform(-1, 1, (B, s.vec_scalars)).astype(dt)
    hv = rng.uniform(-1, 1, (B, s.vec_scalars)).astype(dt)
    Hs = np.stack([rs.transform_unordered(hv[i], FORWARD) for i in range(B)])
    scaling = 1.0 / N
    xd = torch.from_numpy(x).cuda()
    Hd = torch.from_numpy(Hs).cuda()
    # real N = 32 in double is below the double minimum? (both precisions share the size rules) - every listed size is legal
    want_b = _ref_convolve(rs, x, Hs[:1], scaling)
    want_v = _ref_convolve(rs, x, Hs, scaling)
    scale = lambda w: np.abs(w).max(axis=1, keepdims=True)
    try:
        for var in (0, 120):
            pa.set_variant(var)
            got = s.convolve_batch(xd, Hd[0].contiguous(), scaling=scaling).cpu().numpy()
            assert (np.abs(got - want_b) / scale(want_b)).max() <= tol, ("broadcast", var)
        pa.set_variant(0)
        got = s.convolve_batch(xd, Hd, scaling=scaling).cpu().numpy()
        assert (np.abs(got - want_v) / scale(want_v)).max() <= tol, "per vector"
        # accumulate on top of a previous result, and in place
        acc = torch.from_numpy(want_v.astype(dt)).cuda()
        got = s.convolve_batch(xd, Hd[0].contiguous(), out=acc, scaling=scaling, accumulate=True).cpu().numpy()
        w = want_b + want_v
        assert (np.abs(got - w) / scale(w)).max() <= 2 * tol, "accumulate"
        buf = xd.clone()
        got = s.convolve_batch(buf, Hd[0].contiguous(), out=buf, scaling=scaling).cpu().numpy()
        assert (np.abs(got - want_b) / scale(want_b)).max() <= tol, "in place"
    finally:
        pa.set_variant(0)
    s.close(); rs.close()


@pytest.mark.parametrize("dt,tol,N,B", [(np.float32, 1e-5, 1024, (1 << 18) + 37), (np.float64, 1e-12, 1024, (1 << 17) + 5)])
def test_convolve_batch_c2_c5_shapes(ref, dt, tol, N, B):
    """The C2 / C5 shapes (N = 1024 complex, float and double) on a long ragged batch: persistent workgroups, in-order pull,
    prefetch - 600 sampled vectors against the reference's sequence, and every vector against the three-launch composition."""
    from oracle.ref import FORWARD
    rs = ref.setup(N, pa.COMPLEX, dt)
    s = pa.Setup(N, pa.COMPLEX, dt)
    rng = np.random.default_rng(7)
    h = rng.uniform(-1, 1, 2 * N).astype(dt)
    H = rs.transform_unordered(h, FORWARD)
    tdt = torch.float32 if dt == np.float32 else torch.float64
    xd = torch.rand(B, 2 * N, device="cuda", dtype=tdt) * 2 - 1
    Hd = torch.from_numpy(H).cuda()
    got = s.convolve_batch(xd, Hd, scaling=1.0 / N)
    try:
        pa.set_variant(120)
        comp = s.convolve_batch(xd, Hd, scaling=1.0 / N)
    finally:
        pa.set_variant(0)
    den = comp.abs().amax(dim=1)
    assert float(((got - comp).abs().amax(dim=1) / den).max()) <= tol
    idx = np.unique(np.concatenate(([0, 1, 2, B - 3, B - 2, B - 1], rng.integers(0, B, 600))))
    xs = xd[torch.from_numpy(idx).cuda()].cpu().numpy()
    want = _ref_convolve(rs, xs, H[None, :], 1.0 / N)
    g = got[torch.from_numpy(idx).cuda()].cpu().numpy()
    assert (np.abs(g - want).max(axis=1) / np.abs(want).max(axis=1)).max() <= tol
    s.close(); rs.close()


# ------------------------------------------------------------------ long batches of the in-order families against the reference
def _uniform(shape, seed, tdt):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    return torch.rand(shape, device="cuda", dtype=tdt, generator=g) * 2 - 1


@pytest.mark.parametrize("dt,tr,N", [(np.float32, pa.COMPLEX, 4320), (np.float32, pa.COMPLEX, 8192), (np.float32, pa.COMPLEX, 8640),
                                     (np.float64, pa.COMPLEX, 2160), (np.float64, pa.COMPLEX, 3072), (np.float64, pa.COMPLEX, 4096),
                                     (np.float32, pa.COMPLEX, 480 * 256), (np.float64, pa.COMPLEX, 480 * 128),
                                     (np.float32, pa.REAL, 16384), (np.float32, pa.COMPLEX, 1024)])
def test_in_order_families_on_long_ragged_batches(ref, dt, tr, N):
    """The kernels that pull their groups in order from the {next, done} counter pair run MANY iterations per workgroup here
    (>= 4 x the groups one grid of resident workgroups takes at once, ragged tail): the large Stockham plans switched to in-order
    pulling in round 3 (float n = 4320 / 8192 / 8640, double n = 2160 / 3072 / 4096), tile passes of 60 KiB and more (L = 480
    column / row tiles), the register-tiled family (C3's size) and the headline kernel.  >= 512 sampled vectors against
    oracle/_ref in all four direction / layout combinations (the every-size walks use batches of 2 - 3)."""
    tdt = torch.float32 if dt == np.float32 else torch.float64
    s = pa.Setup(N, tr, dt)
    rs = ref.setup(N, tr, dt)
    vec_bytes = s.vec_scalars * np.dtype(dt).itemsize
    B = max(4 * 256 * 4 + 7, (1 << 29) // vec_bytes + 7)          # >= 4 groups per resident workgroup slot, ragged
    x = _uniform((B, s.vec_scalars), 40 + N % 1000, tdt)
    rng = np.random.default_rng(N)
    idx = sorted({0, 1, 2, B // 2, B - 3, B - 2, B - 1} | set(rng.integers(0, B, 512).tolist()))
    it = torch.tensor(idx, device="cuda")
    xh = x[it].cpu().numpy()
    tol = (1e-5 if dt == np.float32 else 1e-12) if N & (N - 1) == 0 or dt == np.float32 else 2e-7
    for d in (pa.FORWARD, pa.BACKWARD):
        for o in (False, True):
            y = s.transform_batch(x, None, d, o)
            got = y[it].cpu().numpy().astype(np.float64)
            want = rs.batch(xh, d, o)
            err = (np.abs(got - want).max(axis=1) / np.abs(want).max(axis=1)).max()
            assert err <= tol, (N, d, o, err)
            del y
    s.close(); rs.close()


# ------------------------------------------------------------------ real transforms beyond LDS in two sweeps (fft_tile.h RMODE)
@pytest.mark.parametrize("dt,tol", [(np.float32, 1e-5), (np.float64, 1e-12)])
@pytest.mark.parametrize("lg", [16, 17, 18, 19, 20])
def test_real_forward_two_sweeps(ref, dt, tol, lg):
    """Real forward into the canonical half spectrum, N = 2^16 .. 2^20, as TWO tile passes (half spectra split inside the column
    tiles, Hermitian partner bins stored by the row tiles; variant 122 forces the route for every length that splits, 0 = the
    adopted table, 121 = complex transform + pair sweep): against the reference, against the three-sweep route, in place,
    ragged batch, bin 0 = (DC, Nyquist)."""
    N = 1 << lg
    tdt = torch.float32 if dt == np.float32 else torch.float64
    s = pa.Setup(N, pa.REAL, dt)
    rs = ref.setup(N, pa.REAL, dt)
    B = 5
    x = _uniform((B, N), 90 + lg, tdt)
    xh = x.cpu().numpy()
    from oracle.ref import FORWARD
    want = rs.batch(xh, FORWARD, True)
    try:
        outs = {}
        for var in (122, 0, 121):
            pa.set_variant(var)
            y = s.transform_batch(x, None, pa.FORWARD, True)
            got = y.cpu().numpy().astype(np.float64)
            err = (np.abs(got - want).max(axis=1) / np.abs(want).max(axis=1)).max()
            assert err <= tol, (var, lg, err)
            outs[var] = y
            z = x.clone()
            s.transform_batch(z, z, pa.FORWARD, True)
            assert torch.equal(z, y), (var, lg)
            # ordered == zreorder(unordered) bit for bit on every route (the unordered transform of the two-sweep route is its canonical
            # spectrum through a one-sweep permutation: src/pffft_priv_impl.h:1497-1498 runs the same passes for both layouts)
            fu = s.transform_batch(x, None, pa.FORWARD, False)
            assert torch.equal(s.zreorder_batch(fu, None, pa.FORWARD), y), (var, lg)
            want_u = rs.batch(xh, FORWARD, False)
            erru = (np.abs(fu.cpu().numpy().astype(np.float64) - want_u).max(axis=1) / np.abs(want_u).max(axis=1)).max()
            assert erru <= tol, (var, lg, erru)
        den = outs[121].abs().amax(dim=1, keepdim=True)
        assert float(((outs[122] - outs[121]).abs() / den).max()) <= tol
    finally:
        pa.set_variant(0)
    s.close(); rs.close()


# ------------------------------------------------------------------ tile passes on a run-time mixed-radix plan (fft_tileg.h)
@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("N", [291600, 314928, 500000, 524880, 600000])
def test_runtime_tile_plans(ref, dt, N):
    """Sizes with few factors of two and a large odd part (2^4 3^6 5^2 ... 2^6 3 5^5) had no two-pass plan on the register-tiled tile
    lengths (R0 2^b, b >= 3) and ran the five streaming sweeps; the tile passes of fft_tileg.h take any length 2^a 3^b 5^c up to 864
    (486 x 648, 750 x 800 ...).  Complex N and real 2N, four direction x layout combinations against oracle/_ref on a batch that takes
    the tiles in order from the per-XCD counters (more tiles than workgroups), and on two vectors (static XCD-contiguous map, a grid
    rounded up to whole eights); ordered == zreorder(unordered) bit for bit; in place == out of place."""
    from conftest import relerr, tol_for
    from test_gpu_round3 import _check_size
    dtype = np.float32 if dt == "f32" else np.float64
    tdt = torch.float32 if dt == "f32" else torch.float64
    L = pa.lib()
    import ctypes
    lens = (ctypes.c_int * 3)()
    L.pffft_hip_tile_plan.argtypes = [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int * 3]
    assert L.pffft_hip_tile_plan(N, int(dt == "f64"), 1, lens) >= 2, "no tile plan"
    for tr, NN in ((pa.COMPLEX, N), (pa.REAL, 2 * N)):
        _check_size(ref, NN, tr, dt, batch=2)
        s = pa.Setup(NN, tr, dtype)
        rs = ref.setup(NN, tr, dtype)
        B = 24 if dt == "f32" else 12
        x = _uniform((B, s.vec_scalars), 4242 + N % 1000, tdt)
        xh = x[[0, B // 2, B - 1]].cpu().numpy()
        for d in (pa.FORWARD, pa.BACKWARD):
            got_o = s.transform_batch(x, None, d, True)
            e = relerr(got_o[[0, B // 2, B - 1]].cpu().numpy(), rs.batch(xh, d, True))
            assert e <= tol_for(dt, NN), (dt, tr, NN, d, e)
            xi = x.clone()
            assert torch.equal(s.transform_batch(xi, xi, d, True), got_o), "in place != out of place"
        fu = s.transform_batch(x, None, pa.FORWARD, False)
        assert torch.equal(s.zreorder_batch(fu, None, pa.FORWARD), s.transform_batch(x, None, pa.FORWARD, True))
        s.close(); rs.close()


_FORCED = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import pffft_amd as pa
from oracle import ref as oref
R = oref.get()
dt = {dt!r}
dtype = np.float32 if dt == "f32" else np.float64
tol = 1e-5 if dt == "f32" else 2e-7            # (double with factors 3 / 5: the reference's own float-suffixed constants, conftest.tol_for)
N = {N}
import os
want_plan = [int(v.rstrip("g")) for v in os.environ["PFFFT_HIP_TILE_FORCE"].split(",")]
assert pa.tile_plan(N, dt == "f64", 1) == want_plan, (pa.tile_plan(N, dt == "f64", 1), want_plan)   # (ADVICE r04: a plan with a leading "g" length was dropped silently)
s = pa.Setup(N, pa.COMPLEX, dtype); rs = R.setup(N, pa.COMPLEX, dtype)
g = torch.Generator(device="cuda"); g.manual_seed(7)
x = torch.empty((5, 2 * N), device="cuda", dtype=torch.float32 if dt == "f32" else torch.float64).uniform_(-1, 1, generator=g)
xh = x.cpu().numpy()
for d in (pa.FORWARD, pa.BACKWARD):
    for o in (True, False):
        got = s.transform_batch(x, None, d, o).cpu().numpy()
        want = rs.batch(xh, d, o)
        e = max(float(np.abs(got[i].astype(np.float64) - want[i]).max() / np.abs(want[i]).max()) for i in range(5))
        assert e <= tol, (d, o, e)
X = np.fft.fft(xh[:, 0::2].astype(np.float64) + 1j * xh[:, 1::2].astype(np.float64))
got = s.transform_batch(x, None, pa.FORWARD, True).cpu().numpy()
e = float(np.abs((got[:, 0::2] + 1j * got[:, 1::2]) - X).max() / np.abs(X).max())
assert e <= (1e-5 if dt == "f32" else 1e-12), e
print("FORCED-OK")
"""


@pytest.mark.parametrize("dt,N,plan", [("f32", 10800, "108,100"), ("f32", 10800, "60,180"), ("f32", 10800, "270,40"), ("f32", 10800, "120,90"),
                                       ("f32", 11664, "162g,72g"), ("f64", 10800, "135g,80g"), ("f64", 10800, "40,270"), ("f64", 50000, "125g,400g"),
                                       ("f64", 18000, "225,80g")])
def test_runtime_tile_plans_forced_lengths(dt, N, plan):
    """The short tile lengths (128 / 256 / 512-thread workgroups, radices 2 .. 12, ragged last tiles of 2 .. 14 sequences, odd lengths in
    double, a register-tiled pass next to a run-time one) on sizes whose production route is the streaming passes: the plan is forced by
    PFFFT_HIP_TILE_FORCE (read once per process: a child process), values against oracle/_ref and against a float64 DFT."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PFFFT_HIP_TILE_FORCE=plan)
    r = subprocess.run([sys.executable, "-c", _FORCED.format(root=root, dt=dt, N=N)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FORCED-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ------------------------------------------------------------------ HIP graph capture of the device entries
def test_device_entries_replay_from_a_hip_graph(ref):
    """after one call on the setup and stream (lazily built tables, per-stream work buffers) the launch path neither
    allocates nor synchronises, so the device entries can be captured and replayed - an LDS-resident transform, a transform beyond LDS (two
    tile passes through the stream's work buffer), the fused convolution and a few-block pffastconv call; the replay writes the same values
    as the direct calls, and those meet the reference."""
    st = torch.cuda.Stream()
    rng = np.random.default_rng(11)
    h = rng.uniform(-1, 1, 1500).astype(np.float32)
    with torch.cuda.stream(st):
        s1 = pa.Setup(1024, pa.COMPLEX)
        s2 = pa.Setup(1 << 16, pa.COMPLEX)
        fc = pa.FastConv(h, 0, 0)
        x1 = _uniform((300, 2048), 1, torch.float32); y1 = torch.empty_like(x1); c1 = torch.empty_like(x1)
        x2 = _uniform((5, 2 << 16), 2, torch.float32); y2 = torch.empty_like(x2)
        xs = _uniform((300001,), 3, torch.float32); ys = torch.zeros_like(xs)
        H = s1.transform_batch(x1[:1].contiguous(), None, pa.FORWARD, False).reshape(-1).contiguous()

        def work():
            s1.transform_batch(x1, y1, pa.FORWARD, False)
            s2.transform_batch(x2, y2, pa.FORWARD, True)
            s1.convolve_batch(x1, H, out=c1, scaling=1.0 / 1024)
            return fc.apply(xs, True, out=ys)[1]

        n = work()
        torch.cuda.synchronize()
        want = [t.clone() for t in (y1, y2, c1, ys)]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            work()
        for t in (y1, y2, c1, ys):
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        for got, w in zip((y1, y2, c1, ys), want):
            assert torch.equal(got, w)
    yw, nw, _ = ref.fastconv(xs.cpu().numpy(), h, 0, 0, 1)
    assert n == nw and np.abs(ys[:n].cpu().numpy() - yw).max() <= (yw.max() - yw.min()) / 1e5
    rs = ref.setup(1 << 16, pa.COMPLEX, np.float32)
    from oracle.ref import FORWARD
    wantf = rs.batch(x2.cpu().numpy(), FORWARD, True)
    assert (np.abs(y2.cpu().numpy().astype(np.float64) - wantf).max(axis=1) / np.abs(wantf).max(axis=1)).max() <= 1e-5
    rs.close(); s1.close(); s2.close(); fc.close()


# ------------------------------------------------------------------ one value per vector, whatever the launch shape
@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("tr,N", [(pa.COMPLEX, 256), (pa.COMPLEX, 1024), (pa.COMPLEX, 4096), (pa.REAL, 16384), (pa.COMPLEX, 8192), (pa.COMPLEX, 2400),
                                  (pa.COMPLEX, 1 << 16), (pa.COMPLEX, 10800)])
def test_values_do_not_depend_on_the_launch_shape(dt, tr, N):
    """Round 4 chooses between three launch shapes by the number of groups per resident workgroup - one group per workgroup without a counter,
    the same in dispatch order for up to four groups per workgroup, the persistent in-order loop beyond (first groups static, the counter for
    the rest).  A vector's spectrum must not depend on which one ran: the first vectors of a long batch equal, bit for bit, the same vectors
    transformed as a short batch, for the four direction x layout combinations."""
    dtype = np.float32 if dt == "f32" else np.float64
    tdt = torch.float32 if dt == "f32" else torch.float64
    s = pa.Setup(N, tr, dtype)
    vb = s.vec_scalars * np.dtype(dtype).itemsize
    big = max(64, min((160 << 20) // vb, 40000))          # many groups per workgroup for the LDS-resident sizes
    x = _uniform((big, s.vec_scalars), 31 + N % 97, tdt)
    for d in (pa.FORWARD, pa.BACKWARD):
        for o in (True, False):
            full = s.transform_batch(x, None, d, o)
            for k in (1, 7, min(big, 1500)):
                part = s.transform_batch(x[:k].contiguous(), None, d, o)
                assert torch.equal(part, full[:k]), (dt, tr, N, d, o, k)
            tail = s.transform_batch(x[big - 5:].contiguous(), None, d, o)
            assert torch.equal(tail, full[big - 5:]), (dt, tr, N, d, o, "tail")
    s.close()
