"""GPU parity tests (-m gpu), round 2: the cases VERDICT r01 found untested — BASELINE configs[4] at its stated size
(per-GPU 2^20 and the 1-GPU strong-scaling 2^23 in place), host threads sharing one setup through the legacy entries
(include/pffft/pffft.h:102-105), the beyond-LDS setups on long batches / two streams, the batched FIR entry, N up to the
reference's limit 2^26 (src/pffft_priv_impl.h:1069).  Same bars as tests/test_gpu_parity.py: 1e-5 float / 1e-12 double
per transform against oracle/_ref, the reference's own range/1e5 limit for the FIR (tests/test_pffastconv.c:685)."""
import threading

import numpy as np
import pytest

from conftest import relerr, tol_for

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import pffft_amd as pa  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available() or pa.device_count() < 1:
        pytest.fail("GPU tests need a HIP device: the product has no CPU fallback")
    torch.cuda.set_device(0)


def _uniform(shape, seed, dtype=torch.float32):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    x = torch.empty(shape, device="cuda", dtype=dtype)
    x.uniform_(-1.0, 1.0, generator=g)
    return x


# ------------------------------------------------------------------ BASELINE configs[4] at its stated sizes
def test_c5_full_batch_per_gpu(ref):
    """N=1024 complex double forward, batch 2^20 = the per-GPU shard of configs[4] (16 GiB in + 16 GiB out)."""
    N, B = 1024, 1 << 20
    s = pa.Setup(N, pa.COMPLEX, np.float64)
    x = _uniform((B, 2 * N), 5, torch.float64)
    y = s.transform_batch(x, None, pa.FORWARD, False)
    idx = torch.tensor(sorted({0, 1, 7, 8, 9, 63, 64, 65, B // 2, B - 2, B - 1} | set(range(1000, B, 251)))).cuda()
    assert idx.numel() >= 4096                                # >= 4096 sampled transforms
    rs = ref.setup(N, 1, np.float64)
    assert relerr(y[idx].cpu().numpy(), rs.batch(x[idx].cpu().numpy(), 0, False)) <= 1e-12
    # Parseval per transform over the WHOLE batch, in slices (layout independent)
    worst = 0.0
    for lo in range(0, B, 1 << 17):
        ex = (x[lo:lo + (1 << 17)] ** 2).sum(1); ey = (y[lo:lo + (1 << 17)] ** 2).sum(1)
        worst = max(worst, float(((ey - N * ex).abs() / (N * ex)).max()))
    assert worst <= 1e-13
    s.transform_batch(y, y, pa.BACKWARD, False)               # round trip through the inverse, in place
    assert float((y / N - x).abs().max()) <= 1e-13
    del x, y
    torch.cuda.empty_cache()
    s.close(); rs.close()


def test_c5_strong_scaling_one_gpu_in_place(ref):
    """configs[4] strong scaling at 1 GPU: all 2^23 vectors (128 GiB) transformed IN PLACE (legal:
    include/pffft/pffft.h:157) — what `bench.py --config c5 --scaling strong --gpus 1` runs."""
    N, B = 1024, 1 << 23
    free, _ = torch.cuda.mem_get_info()
    if free < (B * 2 * N * 8) + (8 << 30):
        pytest.fail(f"needs 128 GiB + margin of HBM, {free >> 30} GiB free")
    s = pa.Setup(N, pa.COMPLEX, np.float64)
    x = _uniform((B, 2 * N), 55, torch.float64)
    idx = torch.tensor(sorted({0, 1, 2, B // 3, B // 2, B - 2, B - 1} | set(range(12345, B, 2039)))).cuda()
    assert idx.numel() >= 4096                                # >= 4096 sampled transforms
    keep = x[idx].cpu().numpy()
    e_in = (x[: 1 << 16] ** 2).sum(1)
    s.transform_batch(x, x, pa.FORWARD, False)
    rs = ref.setup(N, 1, np.float64)
    assert relerr(x[idx].cpu().numpy(), rs.batch(keep, 0, False)) <= 1e-12
    e_out = (x[: 1 << 16] ** 2).sum(1)
    assert float(((e_out - N * e_in).abs() / (N * e_in)).max()) <= 1e-13
    del x
    torch.cuda.empty_cache()
    s.close(); rs.close()


# ------------------------------------------------------------------ threads sharing one setup (legacy entries)
@pytest.mark.parametrize("N,tr,dtype", [(1024, pa.COMPLEX, np.float32), (4096, pa.REAL, np.float32),
                                        (1024, pa.COMPLEX, np.float64),
                                        # beyond LDS: two tile passes through the setup's per-stream work buffers (big_mu)
                                        (61440, pa.COMPLEX, np.float32), (28800, pa.REAL, np.float32)])
def test_eight_host_threads_share_one_setup(ref, N, tr, dtype):
    """include/pffft/pffft.h:102-105: a PFFFT_Setup is read-only and may be used by several threads at once, each
    with its own buffers.  8 threads x 100 pffft_transform / _ordered calls (host pointers) on ONE setup, every result
    compared with the reference's."""
    s = pa.Setup(N, tr, dtype)
    rs = ref.setup(N, tr, dtype)
    T, reps = 8, 100
    rng = np.random.default_rng(11)
    xs = rng.uniform(-1, 1, (T, 4, s.vec_scalars)).astype(dtype)
    want_u = np.stack([rs.batch(xs[t], 0, False) for t in range(T)])
    want_o = np.stack([rs.batch(xs[t], 0, True) for t in range(T)])
    tol = 1e-5 if dtype == np.float32 else 1e-12
    errs, exc = [0.0] * T, []

    def work(t):
        try:
            for r in range(reps):
                i = r % 4
                if r & 1:
                    got, want = s.transform_ordered(xs[t, i], pa.FORWARD), want_o[t, i]
                else:
                    got, want = s.transform(xs[t, i], pa.FORWARD), want_u[t, i]
                errs[t] = max(errs[t], relerr(got, want))
        except Exception as e:  # noqa: BLE001
            exc.append(e)

    e0 = pa.error_count()
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not exc, exc
    assert pa.error_count() == e0
    assert max(errs) <= tol, errs
    s.close(); rs.close()


# ------------------------------------------------------------------ beyond-LDS setups: long batches, two streams
def test_big_setup_zconvolve_long_batch(ref):
    """zconvolve_batch on a setup beyond LDS takes the in-order streaming kernel once the batch is
    >= 64 MiB and needs the counter ring the K_BIG branch never allocated.  N = 65536 complex float, batch 160 (80 MiB)."""
    N, B = 65536, 160
    s = pa.Setup(N, pa.COMPLEX)
    assert pa.kernel_name(s) == "fourstep"
    a, b = _uniform((B, 2 * N), 21), _uniform((B, 2 * N), 22)
    ab0 = _uniform((B, 2 * N), 23)
    rs = ref.setup(N, 1)
    for acc in (True, False):
        for bc in (False, True):
            ab = ab0.clone()
            s.zconvolve_batch(a, b, ab, 0.5, accumulate=acc, b_broadcast=bc)
            for i in (0, 1, B // 2, B - 1):
                want = rs.zconvolve(a[i].cpu().numpy(), b[0 if bc else i].cpu().numpy(), ab0[i].cpu().numpy(), 0.5, acc)
                assert relerr(ab[i].cpu().numpy(), want) <= 1e-6, (acc, bc, i)
    s.close(); rs.close()


def test_big_setup_on_two_streams(ref):
    """two streams running the same beyond-LDS setup concurrently must not share scratch."""
    N, B = 65536, 24
    s = pa.Setup(N, pa.COMPLEX)
    xa, xb = _uniform((B, 2 * N), 31), _uniform((B, 2 * N), 32)
    ya_ref = s.transform_batch(xa, None, pa.FORWARD, True).clone()
    yb_ref = s.transform_batch(xb, None, pa.FORWARD, False).clone()
    torch.cuda.synchronize()
    rs = ref.setup(N, 1)
    assert relerr(ya_ref[:2].cpu().numpy(), rs.batch(xa[:2].cpu().numpy(), 0, True)) <= 1e-5
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(6):
        with torch.cuda.stream(sa):
            ya = s.transform_batch(xa, None, pa.FORWARD, True)
        with torch.cuda.stream(sb):
            yb = s.transform_batch(xb, None, pa.FORWARD, False)
        torch.cuda.synchronize()
        assert torch.equal(ya, ya_ref) and torch.equal(yb, yb_ref)
    s.close(); rs.close()


def test_fastconv_unfused_path_on_a_big_block(ref):
    """pffastconv with blockLen 65536 (Nfft/2 = 32768 complex points: beyond LDS, composed path) on a signal of
    > 256 blocks: the zconvolve of that path runs on a K_BIG setup with a long batch."""
    taps, blk = 4096, 65536
    L = 270 * (blk - taps + 1) + 1000
    rng = np.random.default_rng(41)
    x = rng.uniform(-1, 1, L).astype(np.float32)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    yw, nw, bl = ref.fastconv(x, h, blk, 0, 1)
    fc = pa.FastConv(h, blk, 0)
    assert fc.block_len == bl == blk
    y, n = fc.apply(torch.from_numpy(x).cuda(), True)
    assert n == nw
    lim = (yw.max() - yw.min()) / 1e5
    assert np.abs(y.cpu().numpy() - yw).max() <= lim
    fc.close()


# ------------------------------------------------------------------ batched FIR entry
@pytest.mark.parametrize("taps,L,nsig,flags", [(4096, 1 << 20, 6, 0),      # BASELINE configs[3] x 6 signals (fused kernel)
                                              (1024, 300000, 5, 0),        # fused, ragged last block
                                              (4096, 1 << 20, 3, 64),      # PFFASTCONV_CORRELATION
                                              (64, 100001, 7, 0),          # time-domain kernel
                                              (200, 50000, 4, 1),          # complex I/O, two real transforms (td stride 2)
                                              (1500, 40000, 3, 1),         # complex I/O, composed path
                                              (700, 30000, 3, 17)])        # complex I/O, single FFT
@pytest.mark.parametrize("flush", [1, 0])
def test_fastconv_batch_matches_reference_signal_by_signal(ref, taps, L, nsig, flags, flush):
    """pffastconv_hip_apply_batch: every signal of the batch == one pffastconv_apply of the reference on that signal
    (src/pffastconv.c:133-263): same number of outputs (the block schedule is observable), values within the reference
    test's own limit (tests/test_pffastconv.c:685); rows beyond the produced samples are left untouched."""
    cpl = 2 if flags & 1 else 1
    rng = np.random.default_rng(taps + L + nsig)
    xs = rng.uniform(-1, 1, (nsig, cpl * L)).astype(np.float32)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    fc = pa.FastConv(h, 0, flags)
    xd = torch.from_numpy(xs).cuda()
    yd = torch.full_like(xd, 7.0)
    y, n = fc.apply_batch(xd, bool(flush), out=yd)
    torch.cuda.synchronize()
    got = y.cpu().numpy()
    for i in range(nsig):
        yw, nw, _ = ref.fastconv(xs[i], h, 0, flags, flush)
        assert n == nw, (i, n, nw)
        if nw:
            lim = (yw.max() - yw.min()) / 1e5
            assert np.abs(got[i] - yw).max() <= lim, i
    assert bool((yd[:, n * cpl:] == 7.0).all())
    # the single-signal entry produces the same count and (possibly through other internal block lengths) the same values
    y1, n1 = fc.apply(xd[1].contiguous(), bool(flush))
    assert n1 == n
    if n:
        assert float((y1 - y[1]).abs().max()) <= float(y[1].max() - y[1].min()) / 1e5
    fc.close()


def test_fastconv_batch_strided_rows_and_empty():
    """Row strides larger than a signal, and the degenerate batches."""
    taps, L = 4096, 1 << 18
    h = np.random.default_rng(3).uniform(-1, 1, taps).astype(np.float32)
    fc = pa.FastConv(h, 0, 0)
    big = _uniform((4, L + 4096), 77)
    x = big[:, :L]                                # row stride L + 4096
    out = torch.zeros((4, L + 64), device="cuda")
    y, n = fc.apply_batch(x, True, out=out[:, :L])
    for i in range(4):
        yi, ni = fc.apply(x[i].contiguous(), True)
        assert ni == n and float((yi - y[i]).abs().max()) <= float(y[i].max() - y[i].min()) / 1e5
    assert float(out[:, L:].abs().max()) == 0.0
    y0, n0 = fc.apply_batch(x[:0], True)
    assert n0 == n and y0.shape[0] == 0
    fc.close()


# ------------------------------------------------------------------ N up to the reference's limit
@pytest.mark.parametrize("dt,tr,N", [("f32", 1, 1 << 22), ("f32", 0, 1 << 23), ("f64", 1, 1 << 22), ("f32", 1, 1 << 24),
                                     ("f64", 0, 1 << 24), ("f32", 1, 3 << 21), ("f32", 0, 5 << 20)])
def test_sizes_2p22_to_2p24_against_reference(ref, dt, tr, N):
    """src/pffft_priv_impl.h:1069 accepts N up to 2^26; r01 tested the beyond-LDS plans to 2^21 only."""
    dtype = np.float32 if dt == "f32" else np.float64
    tol = 1e-5 if dt == "f32" else (1e-12 if N & (N - 1) == 0 else 2e-7)
    s = pa.Setup(N, tr, dtype)
    rs = ref.setup(N, tr, dtype)
    x = np.random.default_rng(N % 1000).uniform(-1, 1, (2, s.vec_scalars)).astype(dtype)
    xd = torch.from_numpy(x).cuda()
    for ordered in (True, False):
        got = s.transform_batch(xd, None, pa.FORWARD, ordered)
        want = rs.batch(x, 0, ordered)
        assert relerr(got.cpu().numpy(), want) <= tol, ordered
        back = s.transform_batch(got, None, pa.BACKWARD, ordered)
        assert float((back / N - xd).abs().max()) <= (2e-5 if dt == "f32" else 1e-12), ordered
    s.close(); rs.close()


@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_one_2p26_vector_per_precision(ref, dt):
    """The largest N the reference accepts (src/pffft_priv_impl.h:1069): one complex vector per precision, canonical and
    internal layout, in place and out of place."""
    N = 1 << 26
    dtype = np.float32 if dt == "f32" else np.float64
    s = pa.Setup(N, pa.COMPLEX, dtype)
    rs = ref.setup(N, 1, dtype)
    x = np.random.default_rng(26).uniform(-1, 1, 2 * N).astype(dtype)
    xd = torch.from_numpy(x).cuda().reshape(1, -1)
    want_o = rs.transform_ordered(x, 0)
    got = s.transform_batch(xd, None, pa.FORWARD, True)
    tol = 1e-5 if dt == "f32" else 1e-12
    assert relerr(got.cpu().numpy()[0], want_o) <= tol
    gu = s.transform_batch(xd, None, pa.FORWARD, False)
    assert relerr(gu.cpu().numpy()[0], rs.transform_unordered(x, 0)) <= tol
    y = xd.clone()
    s.transform_batch(y, y, pa.FORWARD, True)                 # in place
    assert torch.equal(y, got)
    s.transform_batch(gu, gu, pa.BACKWARD, False)
    assert float((gu / N - xd).abs().max()) <= (4e-5 if dt == "f32" else 1e-11)
    s.close(); rs.close()
    assert pa.Setup  # (2^26 + 16 is rejected like the reference does: tests/test_abi.py)


# ------------------------------------------------------------------ LDS-DMA staged FIR block kernel (fft_dma.h)
@pytest.mark.parametrize("flush", [1, 0])
def test_fastconv_dma_block_kernel_on_a_long_signal(ref, flush):
    """fastconv_dma_kernel (default for Nfft 16384 when a call spans many blocks): 4096 taps over 2^23 samples — dword-aligned
    DMA of overlapping blocks, clamped tail — against the reference over the WHOLE signal."""
    taps, L = 4096, (1 << 23) + 12345
    rng = np.random.default_rng(97)
    x = rng.uniform(-1, 1, L).astype(np.float32)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    yw, nw, _ = ref.fastconv(x, h, 0, 0, flush)
    fc = pa.FastConv(h, 0, 0)
    try:
        pa.set_variant(97)
        y, n = fc.apply(torch.from_numpy(x).cuda(), bool(flush))
        pa.set_variant(0)
        y0, n0 = fc.apply(torch.from_numpy(x).cuda(), bool(flush))      # whatever the default picks
    finally:
        pa.set_variant(0)
    assert n == nw == n0
    lim = (yw.max() - yw.min()) / 1e5
    assert np.abs(y.cpu().numpy() - yw).max() <= lim
    assert np.abs(y0.cpu().numpy() - yw).max() <= lim
    fc.close()


@pytest.mark.parametrize("taps,L,nsig", [(1024, 1 << 22, 2), (600, 3000001, 1), (2048, 1 << 22, 2), (130, 1 << 21, 3),
                                        (3000, 1 << 21, 2), (4096, 1 << 21, 2)])
@pytest.mark.parametrize("flush", [1, 0])
def test_fastconv_partitioned_kernel(ref, taps, L, nsig, flush):
    """fastconv_part_kernel (fft_fir.h): uniformly partitioned overlap-save, one wavefront per 2048-sample block, the pair
    passes and the product with H folded into two coefficients per bin.  Default for <= 2048 taps when a call spans many
    blocks; variant 88 forces it for up to 4096 taps (1 .. 4 partitions).  Same count as the reference's block schedule,
    values within the reference test's limit over the WHOLE signals, samples beyond the produced ones untouched."""
    rng = np.random.default_rng(taps + nsig)
    xs = rng.uniform(-1, 1, (nsig, L)).astype(np.float32)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    fc = pa.FastConv(h, 0, 0)
    xd = torch.from_numpy(xs).cuda()
    yd = torch.full_like(xd, 7.0)
    try:
        pa.set_variant(88)
        y, n = fc.apply_batch(xd, bool(flush), out=yd)
    finally:
        pa.set_variant(0)
    got = y.cpu().numpy()
    for i in range(nsig):
        yw, nw, _ = ref.fastconv(xs[i], h, 0, 0, flush)
        assert n == nw
        assert np.abs(got[i] - yw).max() <= (yw.max() - yw.min()) / 1e5, i
    assert bool((yd[:, n:] == 7.0).all())
    fc.close()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("tr,N", [(pa.COMPLEX, 96), (pa.COMPLEX, 288), (pa.COMPLEX, 640), (pa.COMPLEX, 960), (pa.COMPLEX, 2592),
                                  (pa.COMPLEX, 9216), (pa.REAL, 96), (pa.REAL, 576), (pa.REAL, 1280), (pa.REAL, 1920),
                                  (pa.REAL, 4800), (pa.REAL, 12000)])
def test_stockham_direct_first_stage(ref, dt, tr, N):
    """The direct-first-stage variant of every compile-time Stockham plan (first-stage operands straight from HBM into
    registers, no producer wavefronts / deposit) is adopted per plan from a measured table (stock_df_gen.h); variants 54 / 55
    force it on / off.  Both must meet the parity bar on ragged batches and in place, and - same butterflies on the same
    operands - agree bit for bit."""
    tdt = torch.float32 if dt == np.float32 else torch.float64
    tol = tol_for("f32" if dt == np.float32 else "f64", N)   # (the reference's double build: radix-3/5 constants in float)
    s = pa.Setup(N, tr, dt)
    rs = ref.setup(N, tr, dt)
    try:
        for B in (1, 3, 7, 1031):
            x = _uniform((B, s.vec_scalars), 400 + B, tdt)
            idx = sorted({0, B // 2, B - 1})
            xh = x[idx].cpu().numpy()
            modes = [(pa.FORWARD, False), (pa.FORWARD, True), (pa.BACKWARD, True)]
            for d, o in modes:
                pa.set_variant(54)
                y = s.transform_batch(x, None, d, o)
                z = x.clone(); s.transform_batch(z, z, d, o)
                pa.set_variant(55)
                y0 = s.transform_batch(x, None, d, o)
                pa.set_variant(0)
                yd = s.transform_batch(x, None, d, o)
                assert relerr(y[idx].cpu().numpy(), rs.batch(xh, d, o)) <= tol, (B, d, o)
                assert torch.equal(z, y), (B, d, o)
                assert torch.equal(y0, y), (B, d, o)
                assert torch.equal(yd, y), (B, d, o)
    finally:
        pa.set_variant(0)
    s.close(); rs.close()


@pytest.mark.parametrize("dt,tr,N", [(np.float32, pa.COMPLEX, 96), (np.float32, pa.COMPLEX, 480), (np.float32, pa.COMPLEX, 640),
                                     (np.float32, pa.COMPLEX, 960), (np.float32, pa.COMPLEX, 2000), (np.float32, pa.COMPLEX, 4000),
                                     (np.float32, pa.REAL, 192), (np.float32, pa.REAL, 1280), (np.float32, pa.REAL, 1920),
                                     (np.float32, pa.REAL, 4000), (np.float32, pa.REAL, 8000),
                                     (np.float64, pa.COMPLEX, 160), (np.float64, pa.COMPLEX, 1536), (np.float64, pa.REAL, 480)])
def test_stockham_direct_first_stage_steady_state(ref, dt, tr, N):
    """Long batches: every workgroup runs MANY iterations, so the image ping-pong ACROSS iterations (the direct variant has no
    closing barrier: the next first stage writes the image the last phase does not read) and the split loader / storer
    wavefronts are exercised in steady state.  Direct (54) and producer (55) variants must agree bit for bit over the whole
    batch; a few transforms are checked against the reference."""
    tdt = torch.float32 if dt == np.float32 else torch.float64
    s = pa.Setup(N, tr, dt)
    rs = ref.setup(N, tr, dt)
    vec_bytes = s.vec_scalars * (4 if dt == np.float32 else 8)
    B = ((4 << 30) if vec_bytes <= 4096 else (1 << 30)) // vec_bytes + 3      # ragged tail
    x = _uniform((B, s.vec_scalars), 500 + N, tdt)
    # product build (pffft_hip_has_variants() == 0): selectors 54 / 55 reach the SAME kernel (one variant per plan is instantiated),
    # so the bit-for-bit comparison below compares a kernel with itself - the reference then checks >= 512 sampled vectors
    nsamp = 6 if pa.has_variants() else 512
    idx = sorted({0, 1, B // 3, B // 2, B - 2, B - 1} | set(np.random.default_rng(N).integers(0, B, nsamp).tolist()))
    xh = x[idx].cpu().numpy()
    tol = tol_for("f32" if dt == np.float32 else "f64", N)
    try:
        for d, o in ((pa.FORWARD, False), (pa.FORWARD, True), (pa.BACKWARD, True)):
            pa.set_variant(54)
            y = s.transform_batch(x, None, d, o)
            pa.set_variant(55)
            y0 = s.transform_batch(x, None, d, o)
            pa.set_variant(0)
            assert torch.equal(y0, y), (d, o)
            assert relerr(y[idx].cpu().numpy(), rs.batch(xh, d, o)) <= tol, (d, o)
            del y, y0
    finally:
        pa.set_variant(0)
    s.close(); rs.close()


def _smooth5(m):
    for q in (2, 3, 5):
        while m % q == 0:
            m //= q
    return m == 1


# every size tools/gen_stock_plans.hip instantiates a compile-time plan for (complex points n; real N = 2 n)
_CT_SIZES = [n for n in range(48, 1025, 16) if _smooth5(n) and (n & (n - 1))] + [n for n in range(1040, 10241, 16) if _smooth5(n) and (n & (n - 1))]


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("tr", [pa.COMPLEX, pa.REAL])
def test_every_compile_time_stockham_plan_against_reference(ref, dt, tr):
    """Every generated plan (not only the sizes of the reference's lists): all four direction / layout combinations of a
    ragged batch against the reference, and the direct-first-stage variant bit-identical to the producer variant wherever
    the size runs on the Stockham kernels."""
    dtype = np.float32 if dt == "f32" else np.float64
    tdt = torch.float32 if dt == "f32" else torch.float64
    try:
        for n in _CT_SIZES:
            N = n if tr == pa.COMPLEX else 2 * n
            s = pa.Setup(N, tr, dtype)
            rs = ref.setup(N, tr, dtype)
            tol = tol_for(dt, N)
            x = _uniform((7, s.vec_scalars), 600 + n, tdt)
            xl = _uniform((2051, s.vec_scalars), 601 + n, tdt)
            xh = x.cpu().numpy()
            for d in (pa.FORWARD, pa.BACKWARD):
                for o in (False, True):
                    got = s.transform_batch(x, None, d, o).cpu().numpy()
                    assert relerr(got, rs.batch(xh, d, o)) <= tol, (dt, tr, N, d, o)
                    if "stock" in pa.kernel_name(s):
                        pa.set_variant(54); a = s.transform_batch(xl, None, d, o)
                        pa.set_variant(55); b = s.transform_batch(xl, None, d, o)
                        pa.set_variant(0)
                        assert torch.equal(a, b), (dt, tr, N, d, o)
            s.close(); rs.close()
    finally:
        pa.set_variant(0)


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("tr,N", [(pa.COMPLEX, 11664), (pa.COMPLEX, 50000), (pa.COMPLEX, 58320), (pa.COMPLEX, 172800),
                                  (pa.COMPLEX, 200000), (pa.COMPLEX, 12000), (pa.COMPLEX, 20480), (pa.REAL, 23328),
                                  (pa.REAL, 100000), (pa.REAL, 345600), (pa.REAL, 19200), (pa.REAL, 40960)])
def test_streaming_passes_for_sizes_beyond_the_stockham_plans(ref, dt, tr, N):
    """n = R x N2 with R in registers (now also 9, 25, 27: Cooley-Tukey with constant twiddles) and the rows on a fast kernel:
    the route of every non-power-of-two size beyond the Stockham plans, including the sizes with ONE image in LDS that used to
    run the in-place radix 2-5 kernel.  All directions / layouts against the reference, ragged batch, in place."""
    dtype = np.float32 if dt == "f32" else np.float64
    tdt = torch.float32 if dt == "f32" else torch.float64
    s = pa.Setup(N, tr, dtype)
    rs = ref.setup(N, tr, dtype)
    tol = tol_for(dt, N)                                  # north_star's bar, flat: 1e-5 float whatever N
    x = _uniform((3, s.vec_scalars), 700 + N % 997, tdt)
    xh = x.cpu().numpy()
    for d in (pa.FORWARD, pa.BACKWARD):
        for o in (False, True):
            got = s.transform_batch(x, None, d, o)
            assert relerr(got.cpu().numpy(), rs.batch(xh, d, o)) <= tol, (dt, tr, N, d, o)
            z = x.clone(); s.transform_batch(z, z, d, o)
            assert torch.equal(z, got), (dt, tr, N, d, o)
    s.close(); rs.close()
