"""GPU parity tests (-m gpu): the HIP path, reached through the C ABI (libpffft_hip.so), against
  * the real reference compiled from its own sources (oracle/_ref, shipped prebuilt to the GPU box),
  * the committed golden fixtures generated from it (tests/golden),
  * the numpy restatement (oracle/pffft_oracle.py),
on seeded inputs at small sizes, and through size-independent properties at BASELINE's full sizes.
Bars (BASELINE.json north_star): 1e-5 relative float, 1e-12 double (see conftest.tol_for for the one
documented exception where the reference's own double build is only ~1e-8 accurate)."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES, gkey, relerr, tol_for

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import pffft_amd as pa  # noqa: E402
from oracle import pffft_oracle as po  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available() or pa.device_count() < 1:
        pytest.fail("GPU tests need a HIP device: the product has no CPU fallback")  # fail loudly, never skip silently
    torch.cuda.set_device(0)


def _dt(dt):
    return np.float32 if dt == "f32" else np.float64


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ------------------------------------------------------------------ golden fixtures
@pytest.mark.parametrize("dt,tr,N", GOLDEN_CASES)
def test_against_golden(golden, dt, tr, N):
    k, dtype, tol = gkey(dt, tr, N), _dt(dt), tol_for(dt, N)
    s = pa.Setup(N, tr, dtype)
    x = golden[k + "_x"]
    for ordered, name in ((True, "ordered"), (False, "unordered")):
        fwd = s.transform_batch(_dev(x[None]), None, pa.FORWARD, ordered).cpu().numpy()[0]
        assert relerr(fwd, golden[k + "_fwd_" + name]) <= tol, (name, "fwd")
        bwd = s.transform_batch(_dev(golden[k + "_fwd_" + name][None]), None, pa.BACKWARD, ordered).cpu().numpy()[0]
        assert relerr(bwd, golden[k + "_bwd_" + name]) <= tol, (name, "bwd")
    if N <= 1024:
        fu = golden[k + "_fwd_unordered"]
        zr = s.zreorder_batch(_dev(fu[None]), None, pa.FORWARD).cpu().numpy()[0]
        assert np.array_equal(zr, fu[golden[k + "_perm"]])  # pure permutation: bit exact
        back = s.zreorder_batch(_dev(zr[None]), None, pa.BACKWARD).cpu().numpy()[0]
        assert np.array_equal(back, fu)
        for acc, nm in ((True, "_zc_accumulate"), (False, "_zc_no_accu")):
            ab = _dev(golden[k + "_zc_acc0"][None].copy())
            s.zconvolve_batch(_dev(fu[None]), _dev(golden[k + "_zc_b"][None]), ab, 0.25, acc)
            assert relerr(ab.cpu().numpy()[0], golden[k + nm]) <= (1e-6 if dt == "f32" else 1e-14)
    s.close()


# ------------------------------------------------------------------ the real reference, more sizes, batches
SIZES_C = [16, 32, 48, 64, 80, 96, 128, 160, 192, 240, 256, 288, 384, 480, 512, 576, 640, 800, 864, 1024, 2048,
           2592, 4000, 4096, 8192, 12000, 16384]
SIZES_R = [32, 64, 96, 128, 160, 192, 256, 288, 384, 480, 512, 576, 640, 800, 864, 1024, 2048, 4000, 4096, 8192,
           12000, 16384, 36864]


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("tr", [0, 1])
def test_against_reference_all_sizes(ref, dt, tr):
    dtype = _dt(dt)
    rng = np.random.default_rng(11 + tr)
    sizes = SIZES_C if tr == 1 else SIZES_R
    if dt == "f64":
        sizes = [n for n in sizes if n * (2 if tr == 1 else 1) * 8 <= 160 * 1024]  # LDS-resident limit
    for N in sizes:
        rs = ref.setup(N, tr, dtype)
        s = pa.Setup(N, tr, dtype)
        batch = 5  # odd, not a multiple of the per-workgroup transform count
        x = rng.uniform(-1, 1, (batch, s.vec_scalars)).astype(dtype)
        tol = tol_for(dt, N)
        for ordered in (False, True):
            want = rs.batch(x, 0, ordered)
            got = s.transform_batch(_dev(x), None, pa.FORWARD, ordered).cpu().numpy()
            assert relerr(got, want) <= tol, (dt, tr, N, ordered, "fwd")
            wb = rs.batch(want, 1, ordered)
            gb = s.transform_batch(_dev(want), None, pa.BACKWARD, ordered).cpu().numpy()
            assert relerr(gb, wb) <= tol, (dt, tr, N, ordered, "bwd")
        # in-place == out-of-place bit-exactly (benchmarks/bench_pffft.c:343-349)
        buf = _dev(x)
        oop = s.transform_batch(buf, None, pa.FORWARD, False).cpu().numpy()
        s.transform_batch(buf, buf, pa.FORWARD, False)
        assert np.array_equal(buf.cpu().numpy(), oop)
        s.close(); rs.close()


@pytest.mark.parametrize("dt,tr,N", [("f64", 1, 96), ("f64", 1, 4000), ("f64", 0, 96), ("f64", 0, 4000),
                                     ("f64", 1, 1024), ("f32", 1, 1024), ("f32", 0, 16384)])
def test_against_float64_dft(dt, tr, N):
    """Independent truth: numpy float64 FFT.  This is where the 1e-12 double bar holds for radix-3/5 sizes."""
    dtype = _dt(dt)
    rng = np.random.default_rng(5)
    s = pa.Setup(N, tr, dtype)
    x = rng.uniform(-1, 1, (2, s.vec_scalars)).astype(dtype)
    got = s.transform_batch(_dev(x), None, pa.FORWARD, True).cpu().numpy().astype(np.float64)
    for i in range(2):
        if tr == 1:
            want = np.fft.fft(x[i, 0::2].astype(np.float64) + 1j * x[i, 1::2])
            g = got[i, 0::2] + 1j * got[i, 1::2]
        else:
            full = np.fft.rfft(x[i].astype(np.float64))
            want = full[:-1].copy(); want[0] = full[0].real + 1j * full[-1].real
            g = got[i, 0::2] + 1j * got[i, 1::2]
        assert np.abs(g - want).max() / np.abs(want).max() <= (1e-12 if dt == "f64" else 1e-5 / 4)
    s.close()


def test_numpy_restatement_agrees():
    rng = np.random.default_rng(3)
    for N, tr in ((1024, 1), (64, 0), (480, 0), (2592, 1)):
        s = pa.Setup(N, tr)
        x = rng.uniform(-1, 1, s.vec_scalars).astype(np.float32)
        got = s.transform_batch(_dev(x[None]), None, pa.FORWARD, False).cpu().numpy()[0]
        assert relerr(got, po.transform(x, N, tr, po.FORWARD, False)) <= 1e-5
        s.close()


# ------------------------------------------------------------------ legacy single-vector entries (host pointers)
@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_legacy_host_pointer_api(ref, dt):
    dtype = _dt(dt)
    rng = np.random.default_rng(21)
    for N, tr in ((64, 0), (1024, 1), (96, 1), (1024, 0)):
        rs, s = ref.setup(N, tr, dtype), pa.Setup(N, tr, dtype)
        tol = tol_for(dt, N)
        x = rng.uniform(-1, 1, s.vec_scalars).astype(dtype)
        fu = s.transform(x, pa.FORWARD)
        fo = s.transform_ordered(x, pa.FORWARD)
        assert relerr(fu, rs.transform_unordered(x, 0)) <= tol
        assert relerr(fo, rs.transform_ordered(x, 0)) <= tol
        assert np.array_equal(s.zreorder(fu, pa.FORWARD), fo)       # transform_ordered == transform + zreorder
        assert relerr(s.transform(fu, pa.BACKWARD) / N, x) <= 20 * tol
        # aliasing: input == output (include/pffft/pffft.h:157)
        buf = pa.api._aligned_empty(s.vec_scalars, dtype); buf[:] = x
        s.transform_inplace(buf, pa.FORWARD, ordered=False)
        assert np.array_equal(buf, fu)
        # zconvolve with all three operands aliased (include/pffft/pffft.h:194)
        ab = s.zconvolve(fu, fu, np.zeros_like(fu), 0.5, accumulate=True)
        assert relerr(ab, rs.zconvolve(fu, fu, np.zeros_like(fu), 0.5, True)) <= 10 * tol
        s.close(); rs.close()


# ------------------------------------------------------------------ reference's own generative checks on the HIP path
@pytest.mark.parametrize("N", [32, 64, 1024, 4096, 16384])
@pytest.mark.parametrize("cplx", [0, 1])
def test_single_tone(N, cplx):
    """tests/test_pffft.c:109-247 — dynamic range >= 140 dB, magnitude error <= 1e-6*N-ish, round trip."""
    if cplx == 0 and N < 32:
        return
    s = pa.Setup(N, pa.COMPLEX if cplx else pa.REAL)
    n = np.arange(N)
    for kk in (0, N // 16, 5 * N // 16):
        amp, phi0 = 1.1, np.pi / 8
        if cplx:
            z = amp * np.exp(1j * (2 * np.pi * kk * n / N + phi0))
            x = np.empty(2 * N, np.float32); x[0::2], x[1::2] = z.real, z.imag
            expected = amp * N
        else:
            x = (amp * np.cos(2 * np.pi * kk * n / N + phi0)).astype(np.float32)
            expected = amp * np.cos(phi0) * N if kk == 0 else amp * N / 2
        X = s.transform_batch(_dev(x[None]), None, pa.FORWARD, True).cpu().numpy()[0].astype(np.float64)
        P = X[0::2] ** 2 + X[1::2] ** 2
        if not cplx and kk == 0:
            P[0] = X[0] ** 2
        others = np.delete(P, kk)
        assert 10 * np.log10(P[kk] / max(others.max(), 1e-300)) >= 140.0   # tests/test_pffft.c:57-61
        assert abs(np.sqrt(P[kk]) / abs(expected) - 1) <= 1e-6 * 4          # :67,207-213
        back = s.transform_batch(_dev(X.astype(np.float32)[None]), None, pa.BACKWARD, True).cpu().numpy()[0]
        assert np.sum((back / N - x) ** 2) <= N * 1e-7                     # :229-243
    s.close()


# ------------------------------------------------------------------ full-size properties (BASELINE configs)
def _hash_uniform(shape, seed, device):
    g = torch.Generator(device=device); g.manual_seed(seed)
    return torch.rand(shape, device=device, generator=g) * 2 - 1


def test_c2_full_batch_properties(ref):
    """BASELINE configs[1]: N=1024 complex float, batch 2^20 (8 GiB in / 8 GiB out)."""
    N, B = 1024, 1 << 20
    s = pa.Setup(N, pa.COMPLEX)
    assert pa.kernel_name(s) == "c1024_f32"
    x = _hash_uniform((B, 2 * N), 2, "cuda")
    y = s.transform_batch(x, None, pa.FORWARD, False)
    # (1) sampled transforms against the reference
    # >= 4096 sampled transforms against the reference (edges of the per-workgroup groups + a stride
    # coprime to every group size over the whole batch)
    idx = torch.tensor(sorted({0, 1, 7, 8, 63, 64, 4095, 4096, B // 2, B - 2, B - 1} | set(range(5, B, 251))))
    assert idx.numel() >= 4096
    rs = ref.setup(N, 1)
    assert relerr(y[idx.cuda()].cpu().numpy(), rs.batch(x[idx.cuda()].cpu().numpy(), 0, False)) <= 1e-5
    # (2) Parseval per transform (layout independent): sum |X|^2 = N sum |x|^2
    ex = (x.double() ** 2).sum(dim=1); ey = (y.double() ** 2).sum(dim=1)
    assert float(((ey - N * ex).abs() / (N * ex)).max()) <= 1e-5
    # (3) X[0] (internal index 0 / 4) = sum of the inputs
    assert float((y[:, 0].double() - x[:, 0::2].double().sum(1)).abs().max()) <= 1e-3
    assert float((y[:, 4].double() - x[:, 1::2].double().sum(1)).abs().max()) <= 1e-3
    # (4) round trip through the inverse, in place
    s.transform_batch(y, y, pa.BACKWARD, False)
    assert float((y / N - x).abs().max()) <= 2e-6
    # (5) ordered == zreorder(unordered), bit exact, on a slice
    sl = x[: 1 << 14]
    yo = s.transform_batch(sl, None, pa.FORWARD, True)
    yu = s.transform_batch(sl, None, pa.FORWARD, False)
    assert torch.equal(s.zreorder_batch(yu, None, pa.FORWARD), yo)
    del x, y
    torch.cuda.empty_cache()
    s.close(); rs.close()


def test_c3_full_batch_properties(ref):
    """BASELINE configs[2]: N=16384 real float forward, batch 2^16 (4 GiB)."""
    N, B = 16384, 1 << 16
    s = pa.Setup(N, pa.REAL)
    x = _hash_uniform((B, N), 3, "cuda")
    y = s.transform_batch(x, None, pa.FORWARD, False)
    idx = torch.tensor(sorted({0, 1, 2, 3, B // 3, B - 2, B - 1} | set(range(5, B, 15)))).cuda()   # >= 4096 sampled transforms
    assert idx.numel() >= 4096
    rs = ref.setup(N, 0)
    assert relerr(y[idx].cpu().numpy(), rs.batch(x[idx].cpu().numpy(), 0, False)) <= 1e-5
    # Parseval for the half spectrum: 2*sum|X_k|^2 - DC^2 - Nyq^2 = N sum x^2 (DC at internal 0, Nyquist at 4)
    ex = (x.double() ** 2).sum(1)
    ey = 2 * (y.double() ** 2).sum(1) - y[:, 0].double() ** 2 - y[:, 4].double() ** 2
    assert float(((ey - N * ex).abs() / (N * ex)).max()) <= 1e-5
    s.transform_batch(y, y, pa.BACKWARD, False)
    assert float((y / N - x).abs().max()) <= 5e-6
    del x, y
    torch.cuda.empty_cache()
    s.close(); rs.close()


def test_c5_double_properties(ref):
    """BASELINE configs[4] per-GPU shape, reduced batch for the test: N=1024 complex double."""
    N, B = 1024, 1 << 16
    s = pa.Setup(N, pa.COMPLEX, np.float64)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x = torch.rand((B, 2 * N), device="cuda", generator=g, dtype=torch.float64) * 2 - 1
    y = s.transform_batch(x, None, pa.FORWARD, False)
    idx = torch.tensor(sorted({0, 1, B // 2, B - 2, B - 1} | set(range(3, B, 15)))).cuda()   # >= 4096 sampled transforms
    assert idx.numel() >= 4096
    rs = ref.setup(N, 1, np.float64)
    assert relerr(y[idx].cpu().numpy(), rs.batch(x[idx].cpu().numpy(), 0, False)) <= 1e-12
    s.transform_batch(y, y, pa.BACKWARD, False)
    assert float((y / N - x).abs().max()) <= 1e-13
    s.close(); rs.close()


def test_linearity_and_edge_batches():
    N = 1024
    s = pa.Setup(N, pa.COMPLEX)
    a = _hash_uniform((37, 2 * N), 7, "cuda"); b = _hash_uniform((37, 2 * N), 8, "cuda")
    fa, fb = s.transform_batch(a, None, 0, False), s.transform_batch(b, None, 0, False)
    fab = s.transform_batch(2 * a - 3 * b, None, 0, False)
    assert float((fab - (2 * fa - 3 * fb)).abs().max() / fab.abs().max()) <= 1e-5
    # batch sizes around the waves-per-workgroup / grid boundaries, and the empty batch
    for B in (0, 1, 7, 8, 9, 4095, 4097):
        x = _hash_uniform((B, 2 * N), 9, "cuda")
        y = s.transform_batch(x, None, 0, False)
        if B:
            assert torch.equal(y[-1:], s.transform_batch(x[-1:].contiguous(), None, 0, False))
    s.close()


# ------------------------------------------------------------------ fast convolution
@pytest.mark.parametrize("name", ["ramp", "rand", "cplx2", "cplx1", "corr"])
@pytest.mark.parametrize("flush", [0, 1])
def test_fastconv_golden(golden, name, flush):
    k = f"fc_{name}_flush{flush}"
    L, taps, blk, flags, fl, n, bl = [int(v) for v in golden[k + "_meta"]]
    fc = pa.FastConv(golden[k + "_h"], blk, flags)
    assert fc.block_len == bl
    y, produced = fc.apply(golden[k + "_x"], bool(flush))
    assert produced == n
    want = golden[k + "_y"]
    if want.size:
        lim = (want.max() - want.min()) / 1e5   # the reference test's own limit, tests/test_pffastconv.c:685
        assert np.abs(y - want).max() <= lim
    yd, nd = fc.apply(_dev(golden[k + "_x"]), bool(flush))
    assert nd == n and (not want.size or np.abs(yd.cpu().numpy() - want).max() <= lim)
    fc.close()


def test_c4_fir_config(ref):
    """BASELINE configs[3]: signal 2^20, 4096 taps, overlap-save with Nfft 8192 (255 blocks)."""
    L, taps = 1 << 20, 4096
    rng = np.random.default_rng(4)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    for name, x in (("uniform", rng.uniform(-1, 1, L).astype(np.float32)),
                    ("ramp", (np.arange(L) % 4093).astype(np.float32))):   # tests/test_pffastconv.c:539
        if name == "ramp":
            h = np.array([(-1.0, 1.0, 0.5)[j % 3] for j in range(taps)], dtype=np.float32)
        yw, nw, bl = ref.fastconv(x, h, 0, 0, 1)
        fc = pa.FastConv(h, 0, 0)
        assert fc.block_len == bl == 8192
        y, n = fc.apply(_dev(x), True)
        assert n == nw == L - taps + 1
        # the reference test's limit is range/1e5 (tests/test_pffastconv.c:685); with 4096 taps the integer
        # ramp sums to ~1.4e6 where one float ulp is 0.125 > range/1e5, so never ask for less than 1e-6*max|y|
        lim = max((yw.max() - yw.min()) / 1e5, 1e-6 * np.abs(yw).max())
        assert np.abs(y.cpu().numpy() - yw).max() <= lim
        # and the naive FIR truth on a window (tests/test_pffastconv.c:175-213)
        w = slice(12345, 12345 + 3000)
        naive = np.convolve(x[w.start: w.stop + taps - 1].astype(np.float64), h.astype(np.float64), mode="valid")
        assert np.abs(y.cpu().numpy()[w] - naive).max() <= lim
        fc.close()


# ------------------------------------------------------------------ sizes beyond LDS (four-step path)
@pytest.mark.parametrize("dt,tr,N", [("f32", 1, 32768), ("f32", 1, 65536), ("f32", 0, 65536), ("f32", 0, 131072),
                                     ("f32", 1, 30000), ("f32", 0, 120000), ("f64", 1, 16384), ("f64", 1, 65536),
                                     ("f64", 0, 65536), ("f32", 1, 1 << 20),
                                     # n = R x N2 for every register-sized factor R (fft_big.h big_col_kernel<R>): 3, 5, 6, 10, 12, 15, 32,
                                     # and the recursive plans (2^21 = 32 x (8 x 8192))
                                     ("f32", 1, 24576), ("f32", 1, 40960), ("f32", 1, 49152), ("f32", 1, 81920), ("f32", 1, 98304),
                                     ("f32", 1, 122880), ("f32", 1, 262144), ("f32", 0, 491520), ("f32", 1, 1 << 21),
                                     ("f64", 1, 12288), ("f64", 1, 131072), ("f64", 0, 1 << 20)])
def test_large_sizes_against_reference(ref, dt, tr, N):
    dtype = _dt(dt)
    rng = np.random.default_rng(N)
    rs, s = ref.setup(N, tr, dtype), pa.Setup(N, tr, dtype)
    assert pa.kernel_name(s) == "fourstep"
    x = rng.uniform(-1, 1, (2, s.vec_scalars)).astype(dtype)
    tol = tol_for(dt, N)                               # north_star's bar, flat: 1e-5 float whatever N
    for ordered in (False, True):
        want = rs.batch(x, 0, ordered)
        got = s.transform_batch(_dev(x), None, pa.FORWARD, ordered).cpu().numpy()
        assert relerr(got, want) <= tol, (ordered, "fwd")
        wb = rs.batch(want, 1, ordered)
        gb = s.transform_batch(_dev(want), None, pa.BACKWARD, ordered).cpu().numpy()
        assert relerr(gb, wb) <= tol, (ordered, "bwd")
    buf = _dev(x)
    s.transform_batch(buf, buf, pa.FORWARD, True)      # in place
    assert relerr(buf.cpu().numpy(), rs.batch(x, 0, True)) <= tol
    s.close(); rs.close()


# ------------------------------------------------------------------ mixed-radix Stockham kernels (non-power-of-two sizes)
STOCK_C = [48, 96, 160, 240, 480, 640, 800, 960, 1200, 2400, 2592, 4000, 9216]
STOCK_R = [96, 160, 480, 800, 1600, 2400, 4000, 9216, 12000]


@pytest.mark.parametrize("variant", [0, 53, 52, 42])
@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_stockham_paths_against_float64_dft(dt, variant):
    """Every code path of fft_stock.h: compile-time plans (0), run-time plans (53), the workgroup kernel where the
    wave-local one is the default (52), chunked in-order scheduling (42) - against numpy's float64 FFT, with batches
    that need several passes per workgroup, a ragged tail, ordered == zreorder(unordered) bit-exactly, and the
    round trip through both inverse layouts."""
    dtype = _dt(dt)
    tol = 2e-6 if dt == "f32" else 1e-12
    rng = np.random.default_rng(77)
    pa.set_variant(variant)
    try:
        for tr, sizes in ((1, STOCK_C), (0, STOCK_R)):
            if variant != 0:
                sizes = sizes[::3]
            for N in sizes:
                s = pa.Setup(N, tr, dtype)
                if pa.kernel_name(s) != "stockham":   # (double: the two images of the largest sizes exceed LDS)
                    s.close(); continue
                batch = max(3, min(4099, (48 << 20) // (s.vec_scalars * np.dtype(dtype).itemsize)))
                x = rng.uniform(-1, 1, (batch, s.vec_scalars)).astype(dtype)
                xd = _dev(x)
                fo = s.transform_batch(xd, None, pa.FORWARD, True)
                g = fo.cpu().numpy().astype(np.float64)
                if tr == 1:
                    want = np.fft.fft(x[:, 0::2].astype(np.float64) + 1j * x[:, 1::2], axis=1)
                else:
                    full = np.fft.rfft(x.astype(np.float64), axis=1)
                    want = full[:, :-1].copy(); want[:, 0] = full[:, 0].real + 1j * full[:, -1].real
                err = np.abs((g[:, 0::2] + 1j * g[:, 1::2]) - want).max() / np.abs(want).max()
                assert err <= tol, (dt, tr, N, variant, err)
                fu = s.transform_batch(xd, None, pa.FORWARD, False)
                assert torch.equal(s.zreorder_batch(fu, None, pa.FORWARD), fo), (dt, tr, N, variant)
                for spec, ordered in ((fo, True), (fu, False)):
                    back = s.transform_batch(spec, None, pa.BACKWARD, ordered).cpu().numpy().astype(np.float64) / N
                    assert np.abs(back - x).max() <= 4 * tol, (dt, tr, N, variant, ordered)
                s.close()
    finally:
        pa.set_variant(0)


def test_stockham_matches_reference_inplace(ref):
    """In place == out of place bit-exactly (benchmarks/bench_pffft.c:343-349) with prefetching producers running ahead."""
    for N, tr in ((2400, 1), (480, 1), (9216, 0), (800, 0)):
        s, rs = pa.Setup(N, tr, np.float32), ref.setup(N, tr, np.float32)
        x = np.random.default_rng(N).uniform(-1, 1, (1500, s.vec_scalars)).astype(np.float32)
        for d, o in ((pa.FORWARD, False), (pa.FORWARD, True), (pa.BACKWARD, True), (pa.BACKWARD, False)):
            buf = _dev(x)
            oop = s.transform_batch(buf, None, d, o)
            s.transform_batch(buf, buf, d, o)
            assert torch.equal(buf, oop), (N, tr, d, o)
        want = rs.batch(x[:4], 0, False)
        got = s.transform_batch(_dev(x[:4]), None, pa.FORWARD, False).cpu().numpy()
        assert relerr(got, want) <= 1e-5
        s.close(); rs.close()


def test_concurrent_streams_share_a_setup():
    """One setup used from two streams at once (the reference allows sharing a setup between threads,
    include/pffft/pffft.h:102-105): launches of the dynamically scheduled kernels take their work counters
    from a ring, so interleaved launches must not disturb each other."""
    for N, tr in ((1024, 1), (4096, 1), (480, 1), (16384, 0)):
        s = pa.Setup(N, tr, np.float32)
        batch = max(64, (64 << 20) // (s.vec_scalars * 4))
        g = torch.Generator(device="cuda"); g.manual_seed(N)
        xa = torch.rand(batch, s.vec_scalars, device="cuda", generator=g) * 2 - 1
        xb = torch.rand(batch, s.vec_scalars, device="cuda", generator=g) * 2 - 1
        wa = s.transform_batch(xa, None, pa.FORWARD, False).clone()
        wb = s.transform_batch(xb, None, pa.FORWARD, False).clone()
        torch.cuda.synchronize()
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        ya, yb = torch.empty_like(xa), torch.empty_like(xb)
        for _ in range(6):
            with torch.cuda.stream(sa):
                s.transform_batch(xa, ya, pa.FORWARD, False)
            with torch.cuda.stream(sb):
                s.transform_batch(xb, yb, pa.FORWARD, False)
        torch.cuda.synchronize()
        assert torch.equal(ya, wa) and torch.equal(yb, wb), (N, tr)
        s.close()


@pytest.mark.parametrize("variant", [0, 42, 60])
def test_spectrum_helpers_paths(ref, variant):
    """zreorder / zconvolve: LDS-image + streaming kernels (0), their in-order chunk scheduler (42), the direct
    grid-stride kernels (60) - against the reference on ragged batches."""
    pa.set_variant(variant)
    try:
        for dt in ("f32", "f64"):
            dtype = _dt(dt)
            for N, tr in ((96, 1), (1024, 1), (2400, 0), (16384, 0), (64, 0)):
                rs, s = ref.setup(N, tr, dtype), pa.Setup(N, tr, dtype)
                batch = 2051 if N <= 2400 else 131
                rng = np.random.default_rng(N + variant)
                a = rng.uniform(-1, 1, (batch, s.vec_scalars)).astype(dtype)
                b = rng.uniform(-1, 1, (batch, s.vec_scalars)).astype(dtype)
                c = rng.uniform(-1, 1, (batch, s.vec_scalars)).astype(dtype)
                can = s.zreorder_batch(_dev(a), None, pa.FORWARD)
                back = s.zreorder_batch(can, None, pa.BACKWARD)
                assert torch.equal(back.cpu(), torch.from_numpy(a)), (dt, N, tr, variant)
                assert np.array_equal(can[:3].cpu().numpy(), np.stack([rs.zreorder(a[i], 0) for i in range(3)]))
                for acc in (True, False):
                    got = s.zconvolve_batch(_dev(a), _dev(b), _dev(c), 0.37, accumulate=acc).cpu().numpy()
                    for i in (0, batch // 2, batch - 1):
                        want = rs.zconvolve(a[i], b[i], c[i], 0.37, acc)
                        assert relerr(got[i], want) <= tol_for(dt, N), (dt, N, tr, variant, acc, i)
                s.close(); rs.close()
    finally:
        pa.set_variant(0)


@pytest.mark.parametrize("dt,N,tr", [("f32", 1024, 1), ("f32", 96, 1), ("f32", 2048, 0), ("f32", 32, 0), ("f64", 1024, 1),
                                     ("f64", 96, 0)])
def test_zconvolve_long_batches(ref, dt, N, tr):
    """zconvolve on batches long enough (> 64 MiB per stream) for the in-order streaming kernel with the DPP pair
    exchange (fft_aux.h zconvolve_dyn_kernel): accumulate / no_accu, one b per vector / one b for all (the FIR case,
    src/pffastconv.c:238), ragged last chunk, aliased output — against the reference on sampled vectors and,
    bit for bit, against the grid-stride kernel (variant 60) on the whole batch."""
    dtype = _dt(dt)
    rs, s = ref.setup(N, tr, dtype), pa.Setup(N, tr, dtype)
    vb = s.vec_scalars * np.dtype(dtype).itemsize
    batch = (68 << 20) // vb + 3
    g = torch.Generator(device="cuda").manual_seed(N)
    tdt = torch.float32 if dt == "f32" else torch.float64
    a = torch.rand(batch, s.vec_scalars, device="cuda", dtype=tdt, generator=g) * 2 - 1
    b = torch.rand(batch, s.vec_scalars, device="cuda", dtype=tdt, generator=g) * 2 - 1
    c0 = torch.rand(batch, s.vec_scalars, device="cuda", dtype=tdt, generator=g) * 2 - 1
    sel = [0, 1, batch // 3, batch - 2, batch - 1]
    ah, bh, ch = a[sel].cpu().numpy(), b[sel].cpu().numpy(), c0[sel].cpu().numpy()
    for acc in (True, False):
        for bc in (False, True):
            bb = b[:1].contiguous() if bc else b
            got = s.zconvolve_batch(a, bb, c0.clone(), 0.37, accumulate=acc, b_broadcast=bc)
            pa.set_variant(60)
            try:
                want_all = s.zconvolve_batch(a, bb, c0.clone(), 0.37, accumulate=acc, b_broadcast=bc)
            finally:
                pa.set_variant(0)
            # same products, same order of operations: identical up to the contraction-free rounding of both kernels
            assert torch.allclose(got, want_all, rtol=0, atol=(1e-6 if dt == "f32" else 1e-14)), (acc, bc)
            gh = got[sel].cpu().numpy()
            for k, i in enumerate(sel):
                want = rs.zconvolve(ah[k], bh[0] if bc else bh[k], ch[k], 0.37, acc)
                assert relerr(gh[k], want) <= tol_for(dt, N), (dt, N, tr, acc, bc, i)
    # output aliased with an input (include/pffft/pffft.h:194)
    a2 = a.clone()
    s.zconvolve_batch(a2, b, a2, 0.5, accumulate=False)
    want = rs.zconvolve(ah[2], bh[2], np.zeros_like(ah[2]), 0.5, False)
    assert relerr(a2[sel[2]].cpu().numpy(), want) <= tol_for(dt, N)
    s.close(); rs.close()


@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_ordered_is_reordered_unordered_bit_for_bit(dt):
    """pffft_transform_ordered == pffft_zreorder(pffft_transform) EXACTLY, both directions, for every power-of-two size
    and several mixed-radix ones — what the reference's own validation asserts of itself
    (benchmarks/bench_pffft.c:343-349).  The launcher picks kernels per size, layout and direction: this pins the rule
    that both layouts of one direction share their arithmetic."""
    dtype = _dt(dt)
    for tr in (pa.COMPLEX, pa.REAL):
        for N in (32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 96, 480, 2400, 4000, 9216):
            if dt == "f64" and N * (2 if tr == pa.COMPLEX else 1) * 8 > 128 * 1024:
                continue
            if N == 32768 and tr == pa.COMPLEX:
                continue                      # beyond LDS: composed from separate sweeps, not a single-kernel size
            s = pa.Setup(N, tr, dtype)
            x = _dev(np.random.default_rng(N).uniform(-1, 1, (5, s.vec_scalars)).astype(dtype))
            fu = s.transform_batch(x, None, pa.FORWARD, False)
            fo = s.transform_batch(x, None, pa.FORWARD, True)
            assert torch.equal(s.zreorder_batch(fu, None, pa.FORWARD), fo), (dt, tr, N, "forward")
            bu = s.transform_batch(fu, None, pa.BACKWARD, False)
            bo = s.transform_batch(fo, None, pa.BACKWARD, True)
            assert torch.equal(bu, bo), (dt, tr, N, "backward")
            s.close()


def test_partitioned_convolution_with_zconvolve_accumulate():
    """The use pffft_zconvolve_accumulate exists for (reference README.md:273-275): a uniformly
    partitioned overlap-save FIR — spectra of the last P input blocks times the P filter-partition spectra, accumulated
    in the internal layout, ONE inverse transform per output block.  Built only from the drop-in's batched entries
    (real transforms: exercises the DC/Nyquist rule of zconvolve, src/pffft_priv_impl.h:1626-1629) and checked against
    a float64 direct convolution with the reference's FIR bar (tests/test_pffastconv.c:685)."""
    B, P, nblk = 256, 4, 40                       # block = partition length, partitions, signal blocks
    N = 2 * B
    rng = np.random.default_rng(5)
    h = rng.uniform(-1, 1, B * P).astype(np.float32)
    x = rng.uniform(-1, 1, B * nblk).astype(np.float32)
    s = pa.Setup(N, pa.REAL, np.float32)
    # filter partitions, zero padded to N, forward transformed (internal layout), pre-scaled by 1/N
    hp = np.zeros((P, N), np.float32); hp[:, :B] = h.reshape(P, B)
    H = s.transform_batch(_dev(hp), None, pa.FORWARD, ordered=False)
    # overlapping input frames [block b-1 | block b], forward transformed
    xp = np.concatenate([np.zeros(B, np.float32), x])
    frames = np.stack([xp[b * B:b * B + N] for b in range(nblk)])
    X = s.transform_batch(_dev(frames), None, pa.FORWARD, ordered=False)
    acc = torch.zeros_like(X)
    for p in range(P):                            # Y_b += X_{b-p} * H_p for every block at once
        if p == 0:
            s.zconvolve_batch(X, H[0:1].contiguous(), acc, 1.0 / N, accumulate=True, b_broadcast=True)
        else:
            s.zconvolve_batch(X[:nblk - p].contiguous(), H[p:p + 1].contiguous(), acc[p:], 1.0 / N, accumulate=True,
                              b_broadcast=True)
    y = s.transform_batch(acc, None, pa.BACKWARD, ordered=False).cpu().numpy()[:, B:]   # the valid half of every frame
    want = np.convolve(x.astype(np.float64), h.astype(np.float64))[:B * nblk].reshape(nblk, B)
    lim = (want.max() - want.min()) / 1e5
    assert np.abs(y - want).max() < lim
    s.close()


@pytest.mark.parametrize("dt,N,tr", [("f32", 1024, 1), ("f32", 96, 1), ("f32", 16384, 0), ("f32", 32, 0), ("f32", 2400, 0),
                                     ("f64", 1024, 1), ("f64", 96, 0), ("f64", 4096, 1)])
def test_zreorder_long_batches(ref, dt, N, tr):
    """zreorder on batches > 64 MiB: the in-order streaming kernel with next-group prefetch (fft_aux.h zreorder_dyn_kernel)
    is a pure permutation — bit for bit equal to the direct kernel (variant 60) on the whole ragged batch, to the
    reference's own pffft_zreorder on sampled vectors, and its own inverse."""
    dtype = _dt(dt)
    rs, s = ref.setup(N, tr, dtype), pa.Setup(N, tr, dtype)
    vb = s.vec_scalars * np.dtype(dtype).itemsize
    batch = (66 << 20) // vb + 3
    tdt = torch.float32 if dt == "f32" else torch.float64
    x = torch.rand(batch, s.vec_scalars, device="cuda", dtype=tdt, generator=torch.Generator(device="cuda").manual_seed(N)) * 2 - 1
    for d in (pa.FORWARD, pa.BACKWARD):
        got = s.zreorder_batch(x, None, d)
        pa.set_variant(60)
        try:
            want = s.zreorder_batch(x, None, d)
        finally:
            pa.set_variant(0)
        assert torch.equal(got, want), (dt, N, tr, d)
        for i in (0, batch // 2, batch - 1):
            assert np.array_equal(got[i].cpu().numpy(), rs.zreorder(x[i].cpu().numpy(), d)), (dt, N, tr, d, i)
        assert torch.equal(s.zreorder_batch(got, None, 1 - d), x)
    s.close(); rs.close()


@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_multiwave_sizes_edge_batches_and_inplace(ref, dt):
    """The sizes whose kernels are picked per layout and direction (three-stage float 2048..8192, TiledAltF64 double
    128..4096): batches of 1, 2, 3, 9 and 17 vectors (fewer / more than one workgroup pass), every direction and layout,
    in place == out of place bit for bit, against the reference."""
    dtype = _dt(dt)
    sizes = [(2048, 1), (4096, 1), (8192, 1), (4096, 0), (8192, 0), (16384, 0)] if dt == "f32" else \
            [(128, 1), (256, 1), (512, 1), (1024, 1), (2048, 1), (4096, 1), (256, 0), (512, 0), (1024, 0), (2048, 0), (4096, 0), (8192, 0)]
    rng = np.random.default_rng(3)
    for N, tr in sizes:
        rs, s = ref.setup(N, tr, dtype), pa.Setup(N, tr, dtype)
        tol = tol_for(dt, N)
        for batch in (1, 2, 3, 9, 17):
            x = rng.uniform(-1, 1, (batch, s.vec_scalars)).astype(dtype)
            for ordered in (False, True):
                want = rs.batch(x, 0, ordered)
                got = s.transform_batch(_dev(x), None, pa.FORWARD, ordered)
                assert relerr(got.cpu().numpy(), want) <= tol, (dt, tr, N, batch, ordered, "fwd")
                buf = _dev(x)
                s.transform_batch(buf, buf, pa.FORWARD, ordered)
                assert torch.equal(buf, got), (dt, tr, N, batch, ordered, "fwd in place")
                wb = rs.batch(want, 1, ordered)
                gb = s.transform_batch(_dev(want), None, pa.BACKWARD, ordered)
                assert relerr(gb.cpu().numpy(), wb) <= tol, (dt, tr, N, batch, ordered, "bwd")
                buf = _dev(want)
                s.transform_batch(buf, buf, pa.BACKWARD, ordered)
                assert torch.equal(buf, gb), (dt, tr, N, batch, ordered, "bwd in place")
        s.close(); rs.close()


@pytest.mark.parametrize("taps", [1, 2, 7, 8, 9, 31, 64, 100, 255, 256, 257, 511, 512, 513])
def test_fastconv_short_filters(ref, taps):
    """pffastconv with short filters (<= 512 taps take the time-domain kernel, 513 the FFT path): same number of samples
    as the reference for flush / no flush (the block schedule is observable, src/pffastconv.c:156-166,204-210), values
    against the reference within its own test limit (tests/test_pffastconv.c:685) and against the direct float64 sum
    y[m] = sum_i x[m+i] hrev[i] (tests/test_pffastconv.c:175-213); convolution and PFFASTCONV_CORRELATION; signal lengths
    that end inside a tile / a block; device and host pointers."""
    rng = np.random.default_rng(taps)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    for L in (taps, taps + 5, 2048 + taps - 1, 5000, 70001):
        x = rng.uniform(-1, 1, L).astype(np.float32)
        for flags in (0, 64):                       # 64 = PFFASTCONV_CORRELATION
            for flush in (1, 0):
                yw, nw, bl = ref.fastconv(x, h, 0, flags, flush)
                fc = pa.FastConv(h, 0, flags)
                assert fc.block_len == bl
                yd, nd = fc.apply(_dev(x), bool(flush))
                assert nd == nw, (taps, L, flags, flush)
                if nw:
                    hrev = h if flags else h[::-1]
                    truth = np.correlate(x.astype(np.float64), hrev.astype(np.float64), "valid")[:nw]
                    lim = max((truth.max() - truth.min()) / 1e5, 2e-6 * np.sqrt(taps) * max(1.0, np.abs(truth).max()))
                    got = yd.cpu().numpy()
                    assert np.abs(got - truth).max() <= lim, (taps, L, flags, flush)
                    assert np.abs(got - yw[:nw]).max() <= 2 * lim
                    yh, nh = fc.apply(x, bool(flush))          # host pointers through pffastconv_apply
                    assert nh == nw and np.array_equal(yh, got)
                fc.close()


@pytest.mark.parametrize("taps", [3, 16, 64, 255])
@pytest.mark.parametrize("flags", [1, 1 | 16, 1 | 64, 1 | 16 | 64])
def test_fastconv_short_filters_complex_io(ref, taps, flags):
    """the complex-I/O modes (PFFASTCONV_CPLX_INP_OUT, with and without CPLX_SINGLE_FFT, with CORRELATION) on short
    filters: stride-2 time-domain kernel; produced counts and values against the reference (which runs its FFT route)."""
    rng = np.random.default_rng(taps + flags)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    for L in (taps + 1, 1500, 40000):
        x = rng.uniform(-1, 1, 2 * L).astype(np.float32)
        for flush in (1, 0):
            yw, nw, bl = ref.fastconv(x, h, 0, flags, flush)
            fc = pa.FastConv(h, 0, flags)
            assert fc.block_len == bl
            yd, nd = fc.apply(_dev(x), bool(flush))
            assert nd == nw, (taps, L, flags, flush)
            if nw:
                hrev = h if flags & 64 else h[::-1]
                xc = x[0::2].astype(np.float64) + 1j * x[1::2]
                truth = np.correlate(xc, hrev.astype(np.float64), "valid")[:nw]
                tr = np.empty(2 * nw); tr[0::2], tr[1::2] = truth.real, truth.imag
                lim = max((tr.max() - tr.min()) / 1e5, 2e-6 * np.sqrt(taps) * max(1.0, np.abs(tr).max()))
                got = yd.cpu().numpy()
                assert np.abs(got - tr).max() <= lim, (taps, L, flags, flush)
                assert np.abs(got - yw[:2 * nw]).max() <= 2 * lim
            fc.close()


@pytest.mark.parametrize("taps", [129, 600, 1024, 2048, 4096, 5000])
def test_fastconv_long_signals(ref, taps):
    """Throughput regime (signals long enough for >= one internal block per CU): the same outputs are computed through
    longer INTERNAL blocks (Nfft 8192 / 16384, pffastconv_impl.h fc_big_nfft) while the number of samples a call produces
    follows the reference's block schedule exactly — flush and no flush — and the values meet the reference's FIR bar
    against a float64 direct convolution (sampled windows) and against the reference on the whole signal."""
    L = (1 << 22) + 12345
    rng = np.random.default_rng(taps)
    x = rng.uniform(-1, 1, L).astype(np.float32)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    fc = pa.FastConv(h, 0, 0)
    xd = _dev(x)
    for flush in (1, 0):
        yw, nw, bl = ref.fastconv(x, h, 0, 0, flush)
        assert fc.block_len == bl
        yd, nd = fc.apply(xd, bool(flush))
        assert nd == nw, (taps, flush)
        got = yd.cpu().numpy()
        lim = max((yw.max() - yw.min()) / 1e5, 2e-6 * np.sqrt(taps) * np.abs(yw).max())
        assert np.abs(got - yw[:nw]).max() <= 2 * lim
        for m0 in (0, 7000, nw // 2, nw - 3000):
            seg = x[m0:m0 + 3000 + taps - 1].astype(np.float64)
            truth = np.correlate(seg, h[::-1].astype(np.float64), "valid")[:3000]
            assert np.abs(got[m0:m0 + 3000] - truth).max() <= lim, (taps, flush, m0)
    fc.close()
