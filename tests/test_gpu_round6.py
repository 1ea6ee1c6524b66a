"""Round-6 rows: the 32-points-per-thread FIR block kernel (fft_fir32.h) held to the reference and to the split kernel it replaced; one
setup used from several devices (per-device replicas of the plan's device state) - on a one-GPU box through the test hook that moves a
thread to a device key of its own, on a multi-GPU box for real; the argument matrix of pffft_hip_transform_batch_multi; NaN / Inf / denormal
vectors through both libraries; HIP-graph replays of the LDS-resident in-order routes.  All through the C ABI against oracle/_ref."""
import ctypes as C
import threading

import numpy as np
import pytest

import pffft_amd as pa
from conftest import legal_sizes, relerr

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

AB_FIR_SPLIT, AB_FAKE_DEVICE = 119, 130


@pytest.fixture(scope="module")
def ref():
    from oracle import ref as oref
    if not oref.available():
        from conftest import missing_checker
        missing_checker("oracle/_ref/libpffft_ref.so")
    return oref.get()


def _uniform(shape, seed, tdt=None):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    return torch.rand(shape, device="cuda", dtype=tdt or torch.float32, generator=g) * 2 - 1


# ------------------------------------------------------------------ FIR: 16384-sample blocks on 256 threads (fft_fir32.h)
@pytest.mark.parametrize("taps,nsig,L", [(4096, 1, (1 << 22) + 12345), (4096, 40, 1 << 17), (2048, 1, 1 << 22), (1500, 3, 1500000),
                                         (851, 1, 4000001), (6000, 2, 3000000), (8192, 1, 1 << 22), (4096, 300, 20000)])
def test_fir32_block_kernel_against_reference_and_split_kernel(ref, taps, nsig, L):
    """The throughput regime of pffastconv (> 850 taps, at least one block per CU): one 256-thread workgroup per 16384-sample block, 32
    points per thread, four exchanges (fft_fir32.h) - the same count and samples as the reference's block loop (src/pffastconv.c:207-261;
    limit of tests/test_pffastconv.c:685), as the float64 direct sum on a window, and as the split kernel (variant 119, the second route
    to the same answer: two independent kernels, one result within a few ulp of the output range); nothing written beyond the produced
    samples; signals whose last block is partial or shorter than one row of the gather."""
    rng = np.random.default_rng(taps + nsig)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    x = _uniform((nsig, L), taps * 3 + nsig)
    fc = pa.FastConv(h, 0, 0)
    y = torch.full((nsig, L), 7.0, device="cuda")
    if nsig == 1:
        _, n = fc.apply(x[0], True, out=y[0])
    else:
        _, n = fc.apply_batch(x, True, out=y)
    torch.cuda.synchronize()
    assert n == L - taps + 1
    assert bool((y[:, n:] == 7.0).all()), "samples beyond the produced ones were touched"
    pa.set_variant(AB_FIR_SPLIT)
    try:
        fs = pa.FastConv(h, 0, 0)
        y2 = torch.full((nsig, L), 7.0, device="cuda")
        if nsig == 1:
            _, n2 = fs.apply(x[0], True, out=y2[0])
        else:
            _, n2 = fs.apply_batch(x, True, out=y2)
        torch.cuda.synchronize()
        fs.close()
    finally:
        pa.set_variant(0)
    assert n2 == n
    rng_y = float(y2[:, :n].max() - y2[:, :n].min())
    assert float((y[:, :n] - y2[:, :n]).abs().max()) <= rng_y / 1e5, "fir32 vs split kernel"
    # the reference on the first and the last signal (its block loop walks one signal in ~0.1 s per 2^20 samples)
    for sig in sorted({0, nsig - 1}):
        xs = x[sig].cpu().numpy()
        if L <= (1 << 21):
            yw, nw, _ = ref.fastconv(xs, h, 0, 0, 1)
            assert nw == n
            lim = (yw.max() - yw.min()) / 1e5                     # tests/test_pffastconv.c:685
            assert np.abs(y[sig, :n].cpu().numpy() - yw).max() <= lim, (taps, sig)
        # float64 direct sums (tests/test_pffastconv.c:175-213 slow_conv_R) on windows at the start, across block borders and at the very end
        got = y[sig, :n].cpu().numpy()
        for w0 in (0, 16384 - taps - 50, n // 2, n - 700):
            w0 = max(0, min(w0, n - 700))
            seg = xs[w0:w0 + 700 + taps - 1].astype(np.float64)
            want = np.convolve(seg, h.astype(np.float64), mode="valid")        # the convention of tests/test_gpu_parity.py::test_c4_fir_config
            assert np.abs(got[w0:w0 + 700] - want).max() <= max(np.abs(want).max(), 1.0) * 2e-5, (taps, sig, w0)
    fc.close()


def test_fir32_is_the_route_of_the_c4_throughput_shapes(ref):
    """BASELINE configs[3] in the throughput regime (64 signals of 2^20 samples, 4096 taps): values against the reference's own block loop
    on sampled signals, and stream semantics: two calls on two streams with one setup each give the same samples as one call."""
    rng = np.random.default_rng(44)
    taps, nsig, L = 4096, 64, 1 << 20
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    x = _uniform((nsig, L), 9)
    fc = pa.FastConv(h, 0, 0)
    y, n = fc.apply_batch(x, True)
    torch.cuda.synchronize()
    for sig in (0, 17, 63):
        yw, nw, _ = ref.fastconv(x[sig].cpu().numpy(), h, 0, 0, 1)
        assert nw == n
        assert np.abs(y[sig, :n].cpu().numpy() - yw).max() <= (yw.max() - yw.min()) / 1e5
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    fa, fb = pa.FastConv(h, 0, 0), pa.FastConv(h, 0, 0)
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        ya, _ = fa.apply_batch(x[:32].contiguous(), True)
    with torch.cuda.stream(sb):
        yb, _ = fb.apply_batch(x[32:].contiguous(), True)
    torch.cuda.synchronize()
    assert torch.equal(ya[:, :n], y[:32, :n]) and torch.equal(yb[:, :n], y[32:, :n])
    for f in (fc, fa, fb):
        f.close()


# ------------------------------------------------------------------ one setup, any device
def _mem_free():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


@pytest.mark.parametrize("dt,tr,N", [("f32", pa.COMPLEX, 1024), ("f64", pa.COMPLEX, 1024), ("f32", pa.REAL, 16384), ("f32", pa.COMPLEX, 4000),
                                     ("f32", pa.COMPLEX, 1 << 16), ("f64", pa.REAL, 36864 * 2)])
def test_one_setup_shared_by_threads_on_two_device_keys(ref, dt, tr, N):
    """The reference's setup is immutable and may be shared by concurrent threads (include/pffft/pffft.h:102-105); here it may also be shared by
    threads on DIFFERENT devices: the device state (tables, counter ring, per-stream scratch, staging) is kept per device.  One GPU: thread B
    runs under pffft_hip_set_variant(130), which moves it to a device key of its own - same physical device, a separate replica of the
    state - while thread A uses the setup plainly; both hammer it concurrently (batched and legacy entries, zreorder, zconvolve,
    convolve_batch).  Same values bit for bit, the setup lists two devices, and destroy_setup releases all of it."""
    from conftest import relerr, tol_for
    dtype = np.float32 if dt == "f32" else np.float64
    tdt = torch.float32 if dt == "f32" else torch.float64
    free0 = _mem_free()
    s = pa.Setup(N, tr, dtype)
    assert pa.setup_devices(s) == []
    x = _uniform((37, s.vec_scalars), N % 1000 + 1, tdt)
    want_u = s.transform_batch(x, None, pa.FORWARD, False).clone()
    want_o = s.transform_batch(x, None, pa.FORWARD, True).clone()
    want_b = s.transform_batch(want_u, None, pa.BACKWARD, False).clone()
    want_z = s.zreorder_batch(want_u, None, pa.FORWARD).clone()
    Hs = want_u[0].contiguous()
    want_c = s.convolve_batch(x, Hs, None, 1.0 / N).clone()
    torch.cuda.synchronize()
    assert pa.setup_devices(s) == [0]
    rs = ref.setup(N, tr, dtype)
    assert relerr(want_o[:3].cpu().numpy(), rs.batch(x[:3].cpu().numpy(), pa.FORWARD, True)) <= tol_for(dt, N)
    errs = []

    def work(fake, seed):
        try:
            torch.cuda.set_device(0)
            pa.set_variant(AB_FAKE_DEVICE if fake else 0)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for rep in range(6):
                    k = (37, 5, 20)[rep % 3]
                    xi = x[:k].contiguous()
                    assert torch.equal(s.transform_batch(xi, None, pa.FORWARD, False), want_u[:k]), (fake, rep, "fwd internal")
                    assert torch.equal(s.transform_batch(xi, None, pa.FORWARD, True), want_o[:k]), (fake, rep, "fwd ordered")
                    assert torch.equal(s.transform_batch(want_u[:k].contiguous(), None, pa.BACKWARD, False), want_b[:k]), (fake, rep, "bwd")
                    assert torch.equal(s.zreorder_batch(want_u[:k].contiguous(), None, pa.FORWARD), want_z[:k]), (fake, rep, "zreorder")
                    assert torch.equal(s.convolve_batch(xi, Hs, None, 1.0 / N), want_c[:k]), (fake, rep, "convolve")
                st.synchronize()
            if N <= 16384:                                        # the legacy host-pointer entry on this thread's replica
                xh = x[1].cpu().numpy()
                got = s.transform_ordered(xh, pa.FORWARD)
                assert np.array_equal(got, want_o[1].cpu().numpy()), (fake, "legacy")
        except BaseException as e:                                # noqa: BLE001
            errs.append(repr(e))
        finally:
            pa.set_variant(0)

    ts = [threading.Thread(target=work, args=(i % 2 == 1, i)) for i in range(4)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errs, errs
    assert sorted(pa.setup_devices(s)) == [0, 64], pa.setup_devices(s)
    e0 = pa.error_count()
    s.close(); rs.close()
    del x, want_u, want_o, want_b, want_z, want_c, Hs
    torch.cuda.empty_cache()
    assert pa.error_count() == e0
    # everything the setup held on both device keys is back (the allocator's granularity is 2 MiB)
    assert _mem_free() >= free0 - (8 << 20), (free0, _mem_free())


def test_fastconv_setup_follows_its_user_to_another_device_key(ref):
    """A PFFASTCONV_Setup (not shareable between threads in the reference: include/pffft/pffastconv.h:77-80) holds its filter tables on one
    device and rebuilds them when its user's device changes: the same call from device key 0, key 64, key 0 again - same samples each time."""
    rng = np.random.default_rng(5)
    for taps, L in ((4096, 1 << 20), (2048, 1 << 22), (300, 1 << 20), (17, 200000)):
        h = rng.uniform(-1, 1, taps).astype(np.float32)
        x = _uniform((L,), taps)
        fc = pa.FastConv(h, 0, 0)
        y0, n0 = fc.apply(x, True)
        y0 = y0.clone()
        outs = []

        def other():
            torch.cuda.set_device(0)
            pa.set_variant(AB_FAKE_DEVICE)
            try:
                y1, n1 = fc.apply(x, True)
                torch.cuda.synchronize()
                outs.append((y1.clone(), n1))
            finally:
                pa.set_variant(0)

        t = threading.Thread(target=other); t.start(); t.join()
        y2, n2 = fc.apply(x, True)
        torch.cuda.synchronize()
        assert outs and outs[0][1] == n0 == n2
        rng_y = float(y0[:n0].max() - y0[:n0].min())
        # (the selector of the hook is `any()`: the hooked call may take the route of a non-default selector - same samples to the limit)
        assert float((outs[0][0][:n0] - y0[:n0]).abs().max()) <= rng_y / 1e5
        assert torch.equal(y2[:n0], y0[:n0])
        fc.close()


def _multi(dt):
    L = pa.lib()
    f = L.pffftd_hip_transform_batch_multi if dt == "f64" else L.pffft_hip_transform_batch_multi
    f.restype = C.c_int
    return f


@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_transform_batch_multi_same_setup_and_argument_matrix(ref, dt):
    """pffft[d]_hip_transform_batch_multi (include/pffft_hip.h): THE SAME setup in every slot (round 6); nparts = 0 is a no-op; a part of zero
    vectors in the middle is skipped; a NULL entry in `streams` means that device's default stream; results equal the one-call result bit
    for bit."""
    dtype = np.float64 if dt == "f64" else np.float32
    multi = _multi(dt)
    N, B = 1024, 2500
    x = _uniform((B, 2 * N), 31, torch.float64 if dt == "f64" else torch.float32)
    s = pa.Setup(N, pa.COMPLEX, dtype)
    want = s.transform_batch(x, None, pa.FORWARD, True).clone()
    torch.cuda.synchronize()
    assert multi(0, None, None, None, None, None, pa.FORWARD, 1, None) == 0
    cuts = [0, 900, 900, 901, B]                                  # part 1 is empty
    P = 4
    st = [torch.cuda.Stream(), torch.cuda.Stream(), None, torch.cuda.Stream()]
    y = torch.zeros_like(x)
    devs = (C.c_int * P)(*([0] * P))
    hs = (C.c_void_p * P)(*([s.handle] * P))
    ins = (C.c_void_p * P)(*[x[cuts[i]:].data_ptr() if cuts[i] < B else x.data_ptr() for i in range(P)])
    outs = (C.c_void_p * P)(*[y[cuts[i]:].data_ptr() if cuts[i] < B else y.data_ptr() for i in range(P)])
    bs = (C.c_size_t * P)(*[cuts[i + 1] - cuts[i] for i in range(P)])
    sts = (C.c_void_p * P)(*[q.cuda_stream if q is not None else None for q in st])
    torch.cuda.synchronize()
    assert multi(P, devs, hs, ins, outs, bs, pa.FORWARD, 1, sts) == 0, pa.last_error()
    torch.cuda.synchronize()
    assert torch.equal(y, want)
    y.zero_()
    assert multi(P, devs, hs, ins, outs, bs, pa.FORWARD, 1, None) == 0      # streams == NULL: every part on the default stream
    torch.cuda.synchronize()
    assert torch.equal(y, want)
    assert pa.setup_devices(s) == [0]
    s.close()


@pytest.mark.skipif(pa.device_count() < 2, reason="needs two GPUs (runs by itself on a multi-GPU box)")
@pytest.mark.parametrize("dt,tr,N", [("f32", pa.COMPLEX, 1024), ("f64", pa.COMPLEX, 1024), ("f32", pa.REAL, 1 << 17)])
def test_one_setup_on_every_visible_device(ref, dt, tr, N):
    """Multi-GPU boxes: ONE setup, one thread per device with its own hipSetDevice, then pffft_hip_transform_batch_multi over all visible
    devices with that same setup in every slot - bit-identical with the one-device result; the setup lists every device."""
    from conftest import relerr, tol_for
    dtype = np.float32 if dt == "f32" else np.float64
    tdt = torch.float32 if dt == "f32" else torch.float64
    nd = pa.device_count()
    s = pa.Setup(N, tr, dtype)
    x0 = _uniform((64, s.vec_scalars), 3, tdt)
    want = s.transform_batch(x0, None, pa.FORWARD, False).clone()
    torch.cuda.synchronize()
    rs = ref.setup(N, tr, dtype)
    assert relerr(want[:2].cpu().numpy(), rs.batch(x0[:2].cpu().numpy(), pa.FORWARD, False)) <= tol_for(dt, N)
    errs = []

    def work(d):
        try:
            torch.cuda.set_device(d)
            xd = x0.to(f"cuda:{d}")
            for _ in range(4):
                yd = s.transform_batch(xd, None, pa.FORWARD, False)
                torch.cuda.synchronize(d)
                assert torch.equal(yd.to("cuda:0"), want), d
        except BaseException as e:                                # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(d,)) for d in range(nd)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errs, errs
    assert sorted(pa.setup_devices(s)) == list(range(nd))
    multi = _multi(dt)
    xs = [x0.to(f"cuda:{d}") for d in range(nd)]
    ys = [torch.zeros_like(q) for q in xs]
    devs = (C.c_int * nd)(*range(nd))
    hs = (C.c_void_p * nd)(*([s.handle] * nd))
    ins = (C.c_void_p * nd)(*[q.data_ptr() for q in xs])
    outs = (C.c_void_p * nd)(*[q.data_ptr() for q in ys])
    bs = (C.c_size_t * nd)(*([64] * nd))
    for d in range(nd): torch.cuda.synchronize(d)
    assert multi(nd, devs, hs, ins, outs, bs, pa.FORWARD, 0, None) == 0, pa.last_error()
    for d in range(nd):
        torch.cuda.synchronize(d)
        assert torch.equal(ys[d].to("cuda:0"), want), d
    assert torch.cuda.current_device() == 0
    s.close(); rs.close()


# ------------------------------------------------------------------ NaN / Inf / denormal vectors
@pytest.mark.parametrize("dt,tr,N", [("f32", pa.COMPLEX, 1024), ("f32", pa.REAL, 16384), ("f64", pa.COMPLEX, 1024), ("f32", pa.COMPLEX, 4000),
                                     ("f32", pa.COMPLEX, 1 << 16), ("f64", pa.REAL, 4096)])
def test_nonfinite_and_denormal_vectors_propagate_like_the_reference(ref, dt, tr, N):
    """IEEE special values are data: a NaN or an Inf anywhere in a vector reaches every bin of its ordered spectrum (each output is a sum
    over all inputs) in the reference and here alike - and ONLY that vector's; denormal inputs give the reference's values to the usual bar
    (no flush to zero); none of it counts as a failure (pffft_hip_error_count unchanged: fail-soft NaN output is told apart by the counter)."""
    from conftest import relerr, tol_for
    dtype = np.float32 if dt == "f32" else np.float64
    s, rs = pa.Setup(N, tr, dtype), ref.setup(N, tr, dtype)
    rng = np.random.default_rng(N)
    x = rng.uniform(-1, 1, (6, s.vec_scalars)).astype(dtype)
    tiny = np.finfo(dtype).tiny
    x[1, 5] = np.nan
    x[2, s.vec_scalars // 2 + 1] = np.inf
    x[3, 7] = -np.inf; x[3, 11] = np.inf                          # Inf - Inf somewhere: NaN
    x[4] = (x[4] * tiny / 4).astype(dtype)                        # all denormal
    e0 = pa.error_count()
    for ordered in (True, False):
        got = s.transform_batch(torch.from_numpy(x).cuda(), None, pa.FORWARD, ordered).cpu().numpy()
        want = rs.batch(x, pa.FORWARD, ordered)
        for i in (0, 5):
            assert np.isfinite(got[i]).all() and relerr(got[i], want[i]) <= tol_for(dt, N), (ordered, i)
        for i in (1, 2, 3):
            # every output of the reference is non-finite there.  Ours: every bin, but not necessarily both of its parts - a sample times
            # a twiddle that is exactly +-1 or +-i is a register swap / sign here and a general complex product (NaN x 0 = NaN) in the
            # reference: the few bins k with W^(jk) on an axis keep the part the special value does not feed (8 of 2048 floats at N = 1024)
            bad_w, bad_g = ~np.isfinite(want[i]), ~np.isfinite(got[i])
            # (the reference keeps such parts too where its first pass multiplies by nothing: 16 of 2048 for the Inf at sample 512)
            # (an Inf in the imaginary part of sample N/2 meets W^(k N/2) = +-1 only: exactly the imaginary parts of all bins here)
            assert bad_w.sum() >= 0.98 * bad_w.size and bad_g.sum() >= 0.49 * bad_g.size, (ordered, i, int(bad_w.sum()), int(bad_g.sum()))
            if ordered and tr == pa.COMPLEX:
                assert (bad_g[0::2] | bad_g[1::2]).all(), (i, "a bin with two finite parts")
        den = np.abs(want[4]).max()
        assert den > 0 and np.abs(got[4].astype(np.float64) - want[4]).max() <= max(tol_for(dt, N) * den, 4 * np.finfo(dtype).smallest_subnormal), ordered
    assert pa.error_count() == e0
    s.close(); rs.close()


def test_fir_nonfinite_samples_stay_in_their_blocks(ref):
    """FIR: a NaN sample poisons the outputs whose window holds it in the time domain - and, as in the reference's overlap-save blocks, the
    whole block(s) of the FFT route that hold it; outputs of blocks that do not see the sample are the clean ones."""
    rng = np.random.default_rng(8)
    taps, L = 2048, 1 << 22
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    x = _uniform((L,), 77)
    fc = pa.FastConv(h, 0, 0)
    clean, n = fc.apply(x, True)
    clean = clean.clone()
    pos = 2000000
    xb = x.clone(); xb[pos] = float("nan")
    bad, n2 = fc.apply(xb, True)
    torch.cuda.synchronize()
    assert n2 == n
    nanmask = torch.isnan(bad[:n])
    assert bool(nanmask[pos - taps + 1:pos + 1].all()), "every output whose window holds the NaN"
    lo, hi = pos - 16384 - taps, pos + 16384 + taps                 # at most the blocks around it
    assert not bool(nanmask[:lo].any()) and not bool(nanmask[hi:].any())
    assert torch.equal(bad[:lo], clean[:lo]) and torch.equal(bad[hi:n], clean[hi:n])
    fc.close()


# ------------------------------------------------------------------ HIP-graph replays of the LDS-resident in-order routes
def test_graph_replay_equals_direct_call_on_lds_resident_routes(ref):
    """Captured launches of the persistent in-order kernels (their work counters come from the captured region of the ring): C2 at 2^18
    vectors, the fused convolution and the batched FIR entry replay the values of the direct call bit for bit, three times in a row and
    next to a direct call on another stream."""
    st = torch.cuda.Stream()
    s = pa.Setup(1024, pa.COMPLEX)
    rng = np.random.default_rng(2)
    h = rng.uniform(-1, 1, 4096).astype(np.float32)
    fc = pa.FastConv(h, 0, 0)
    with torch.cuda.stream(st):
        x = _uniform((1 << 18, 2048), 5)
        y = torch.empty_like(x)
        H = s.transform_batch(x[:1].contiguous(), None, pa.FORWARD, False)[0].contiguous()
        yc = torch.empty_like(x)
        xs = _uniform((64, 1 << 20), 6)
        ys = torch.zeros_like(xs)
        s.transform_batch(x, y, pa.FORWARD, False)                # warm-ups: tables on this stream
        s.convolve_batch(x, H, yc, 1.0 / 1024)
        _, n = fc.apply_batch(xs, True, out=ys)
        st.synchronize()
        w_t, w_c, w_f = y.clone(), yc.clone(), ys.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            s.transform_batch(x, y, pa.FORWARD, False)
            s.convolve_batch(x, H, yc, 1.0 / 1024)
            fc.apply_batch(xs, True, out=ys)
        other = torch.cuda.Stream()
        for rep in range(3):
            y.zero_(); yc.zero_(); ys.zero_()
            g.replay()
            with torch.cuda.stream(other):
                z = s.transform_batch(x[:5000].contiguous(), None, pa.FORWARD, False)
            st.synchronize(); other.synchronize()
            assert torch.equal(y, w_t) and torch.equal(yc, w_c) and torch.equal(ys[:, :n], w_f[:, :n]), rep
            assert torch.equal(z, w_t[:5000])
    from conftest import relerr
    rs = ref.setup(1024, pa.COMPLEX, np.float32)
    assert relerr(w_t[:4].cpu().numpy(), rs.batch(x[:4].cpu().numpy(), pa.FORWARD, False)) <= 1e-5
    rs.close(); s.close(); fc.close()


# ------------------------------------------------------------------ bench.py --single-process (one process, N devices, one setup)
@pytest.mark.parametrize("cfg", ["c2", "c5"])
def test_bench_single_process_path(cfg):
    """bench.py --gpus N --single-process: one rank drives N devices through pffft_hip_transform_batch_multi with ONE setup - the JSON line
    of the contract with `devices_seen`; on a 1-GPU box N = 1 for real and N = 2 as the harness test where the parts share the device
    (PFFFT_BENCH_SHARE_GPU=1), on a multi-GPU box every visible device."""
    import json, os, subprocess, sys
    from conftest import ROOT
    nd = pa.device_count()
    for gpus, share in ((nd, False), (2, True)):
        env = dict(os.environ)
        if share:
            env["PFFFT_BENCH_SHARE_GPU"] = "1"
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--single-process", "--config", cfg,
                            "--steps", "3", "--warmup", "1", "--batch-log2", "12"], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == gpus and line["single_process"] and line["steps"] == 3
        assert line["devices_seen"] == (min(gpus, nd) if not share else min(2, nd))
        assert line["parity_vs_reference"]["max_rel_err"] <= (1e-5 if cfg == "c2" else 1e-12)
        assert line["value"] > 0 and line["roofline"]["frac"] > 0


# ------------------------------------------------------------------ the single-image kernel (fft_one.h): one pass for the vectors that fill LDS once
def _oneimage_sizes(tr, dt):
    esz = 8 if dt == "f32" else 16
    out = []
    for N in legal_sizes(tr, 0, 1 << 16):
        n = N if tr == pa.COMPLEX else N // 2
        if 80000 < n * esz <= 147456 and n & (n - 1):
            out.append(N)
    return out


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("tr", [pa.COMPLEX, pa.REAL])
def test_oneimage_every_size_against_reference_and_the_tile_passes(ref, dt, tr):
    """Every size of the single-image kernel, four direction x layout combinations: against the reference's transform (oracle/_ref) at the
    flat bars, against the two / three tile passes the sizes ran on until round 6 (variant 123: an independent route to the same spectrum), in
    place bit-identical to out of place, ordered bit-identical to zreorder(unordered) - on batches that are ragged against the 256 resident
    workgroups (1, 257 and 600 vectors: the static first vector, the in-order counter, its reset between launches)."""
    dtype, tdt, tol = (np.float32, torch.float32, 1e-5) if dt == "f32" else (np.float64, torch.float64, 1e-12)
    sizes = _oneimage_sizes(tr, dt)
    assert len(sizes) == (17 if dt == "f32" else 14), sizes
    rng = np.random.default_rng(606)
    for N in sizes:
        s = pa.Setup(N, tr, dtype)
        assert "oneimage" in pa.describe(s) and pa.kernel_name(s) == "fourstep"
        rs = ref.setup(N, tr, dtype)
        for batch in ((1, 257) if N != sizes[-1] else (1, 257, 600)):
            x = torch.from_numpy(rng.uniform(-1, 1, (batch, s.vec_scalars)).astype(dtype)).cuda()
            pick = sorted({0, batch - 1, batch // 2})
            xh = x[pick].cpu().numpy()
            for d in (pa.FORWARD, pa.BACKWARD):
                res = {}
                for o in (True, False):
                    got = s.transform_batch(x, None, d, o)
                    want = rs.batch(xh, d, o)
                    e = relerr(got[pick].cpu().numpy(), want)
                    lim = tol if (dt == "f32" or all(N % q for q in (3, 5))) else 2e-7     # (the reference's double build: DESIGN.md §4)
                    assert e <= lim, (dt, tr, N, batch, d, o, e)
                    pa.set_variant(123)
                    try:
                        old = s.transform_batch(x, None, d, o)
                    finally:
                        pa.set_variant(0)
                    assert relerr(got[pick].cpu().numpy(), old[pick].cpu().numpy()) <= 4 * tol, (dt, tr, N, batch, d, o)
                    y = x.clone()
                    s.transform_batch(y, y, d, o)
                    assert torch.equal(y, got), (dt, tr, N, batch, d, o, "in place")
                    res[o] = got
                if d == pa.FORWARD:
                    assert torch.equal(s.zreorder_batch(res[False], None, pa.FORWARD), res[True]), (dt, tr, N, batch, "zreorder(unordered) != ordered")
        s.close(); rs.close()


def test_oneimage_streams_graph_and_roundtrip():
    """The single-image kernel on ten streams at once (one work counter per launch), replayed from a HIP graph, and forward -> backward = N x."""
    N, tr = 12000, pa.COMPLEX
    s = pa.Setup(N, tr, np.float32)
    x = torch.rand(700, 2 * N, device="cuda") * 2 - 1
    want = s.transform_batch(x, None, pa.FORWARD, False)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(10)]
    outs = []
    for st in streams:
        with torch.cuda.stream(st):
            outs.append(s.transform_batch(x, None, pa.FORWARD, False))
    torch.cuda.synchronize()
    for y in outs:
        assert torch.equal(y, want)
    st = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    y = torch.empty_like(x)
    with torch.cuda.stream(st):
        s.transform_batch(x, y, pa.FORWARD, False); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            s.transform_batch(x, y, pa.FORWARD, False)
    y.zero_(); g.replay(); g.replay(); torch.cuda.synchronize()
    assert torch.equal(y, want)
    back = s.transform_batch(want, None, pa.BACKWARD, False)
    assert float((back / N - x).abs().max()) <= 2e-5
    s.close()


# ------------------------------------------------------------------ real forward beyond LDS: the pair pass inside the last tile pass (fft_tile.h RMODE 3)
@pytest.mark.parametrize("dt,N", [("f64", 122880), ("f64", 98304), ("f64", 65536), ("f64", 46080), ("f64", 194400), ("f64", 1 << 17),
                                  ("f32", 122880), ("f32", 245760), ("f32", 163840), ("f32", 46080), ("f32", 92160)])
def test_real_forward_pair_pass_inside_the_row_pass(ref, dt, N):
    """Row tiles whose row set is closed under k1 -> N1 - k1 hold both bins of every pair (k, n - k): the pair pass runs in LDS and the tile stores
    the canonical half-complex spectrum - two sweeps instead of three for the ORDERED forward transform (121 = the complex core + pair sweep).  Its
    pair arithmetic is the pair sweeps' operation for operation: the spectrum is BIT-IDENTICAL to the three-sweep route's, and the unordered
    transform - which keeps its three sweeps - still satisfies ordered == zreorder(unordered) bit for bit.  Against the reference, in place,
    batches of 1 and 3 (tile 0 holds rows 0 and N1/2: bin 0 = (DC, Nyquist), bin n/2 = conj Z)."""
    dtype, tdt, tol = (np.float32, torch.float32, 1e-5) if dt == "f32" else (np.float64, torch.float64, 1e-12)
    s = pa.Setup(N, pa.REAL, dtype)
    rs = ref.setup(N, pa.REAL, dtype)
    var = 0 if dt == "f64" else 122          # (adopted in double; variant 122 runs it in float)
    from oracle.ref import FORWARD
    try:
        pa.set_variant(var)
        fused = "real-rows" in pa.describe(s) if var == 0 else True
        for batch in (1, 3):
            x = _uniform((batch, N), 700 + batch, tdt)
            xh = x.cpu().numpy()
            outs = {}
            for v in (var, 121):
                pa.set_variant(v)
                yo = s.transform_batch(x, None, pa.FORWARD, True)
                yu = s.transform_batch(x, None, pa.FORWARD, False)
                lim = tol if (dt == "f32" or all(N % q for q in (3, 5))) else 2e-7      # (the reference's double build: DESIGN.md §4)
                assert relerr(yo.cpu().numpy(), rs.batch(xh, FORWARD, True)) <= lim, (dt, N, v, "ordered")
                assert relerr(yu.cpu().numpy(), rs.batch(xh, FORWARD, False)) <= lim, (dt, N, v, "unordered")
                assert torch.equal(s.zreorder_batch(yu, None, pa.FORWARD), yo), (dt, N, v, "zreorder(unordered) != ordered")
                z = x.clone()
                s.transform_batch(z, z, pa.FORWARD, True)
                assert torch.equal(z, yo), (dt, N, v, "in place")
                outs[v] = yo
            assert torch.equal(outs[var], outs[121]), (dt, N, "the fused pass and the pair sweep differ")
        if N in (122880, 98304, 65536):
            assert fused, pa.describe(s)
    finally:
        pa.set_variant(0)
    s.close(); rs.close()
