"""Round-5 rows: the pffastconv hint flags held to the reference, the scratch-eviction path of a beyond-LDS setup on ten streams, real
transforms and run-time tile plans beyond LDS on long ragged batches, two host threads x two setups that select device 0 themselves,
HIP-graph replays next to direct calls that outgrow the captured scratch, the short-launch kernel of the headline size, first use of one
pffastconv setup from two streams.  All through the C ABI against oracle/_ref."""
import threading

import numpy as np
import pytest

import pffft_amd as pa

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


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


# ------------------------------------------------------------------ pffastconv hint flags (include/pffft/pffastconv.h:100-126)
DIRECT_INP, DIRECT_OUT, SINGLE_FFT, SYMMETRIC, CPLX = 4, 8, 16, 32, 1


@pytest.mark.parametrize("flags,ref_flags", [
    (DIRECT_OUT, DIRECT_OUT), (SYMMETRIC, SYMMETRIC), (DIRECT_OUT | SYMMETRIC, DIRECT_OUT | SYMMETRIC),
    # real DIRECT_INP: the reference's own setup dereferences the input image it did not allocate (src/pffastconv.c:84-85 Xt = NULL,
    # :99 memset(s->Xt ...)) - it cannot run; the flag is a hint ("X may be transformed in place of a copy"), so the values are those
    # of the same call without it
    (DIRECT_INP, 0), (DIRECT_INP | DIRECT_OUT, DIRECT_OUT), (DIRECT_INP | DIRECT_OUT | SYMMETRIC, DIRECT_OUT | SYMMETRIC),
    # complex input / output, two transforms per block: both hints are ignored by the reference (:215 comes first)
    (CPLX | DIRECT_INP, CPLX | DIRECT_INP), (CPLX | DIRECT_OUT, CPLX | DIRECT_OUT), (CPLX | DIRECT_INP | DIRECT_OUT, CPLX | DIRECT_INP | DIRECT_OUT),
    # complex, one transform per block: the reference honours both (:175, :190)
    (CPLX | SINGLE_FFT | DIRECT_INP, CPLX | SINGLE_FFT | DIRECT_INP), (CPLX | SINGLE_FFT | DIRECT_OUT, CPLX | SINGLE_FFT | DIRECT_OUT),
    (CPLX | SINGLE_FFT | DIRECT_INP | DIRECT_OUT | SYMMETRIC, CPLX | SINGLE_FFT | DIRECT_INP | DIRECT_OUT | SYMMETRIC)])
@pytest.mark.parametrize("taps,blk", [(64, 0), (64, 1024), (24, 256), (1000, 0), (4096, 0), (160, 4096)])
def test_fastconv_hint_flags(ref, flags, ref_flags, taps, blk):
    """PFFASTCONV_DIRECT_INP (4), DIRECT_OUT (8), SYMMETRIC (32) are accepted as hints: on inputs that satisfy the reference's conditions
    - one block: inputLen <= blockLen, X and Y of blockLen samples, the tail of X zero, a symmetric filter of 8 k taps - a call returns
    the same count and the same first `count` samples as the reference run with the same flags (what the reference writes beyond them
    with DIRECT_OUT is the wrapped part of its circular convolution: not output).  Device and host entry."""
    rng = np.random.default_rng(taps * 7 + flags)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    h = ((h + h[::-1]) / 2).astype(np.float32)                   # symmetric, taps a multiple of 8
    cpl = 2 if flags & CPLX else 1
    fc = pa.FastConv(h, blk, flags)
    B = fc.block_len
    n_valid = B - 3 if taps < B - 8 else B                       # input shorter than the block, zero tail (condition 4 of DIRECT_INP)
    x = np.zeros(B * cpl, np.float32)
    x[:n_valid * cpl] = rng.uniform(-1, 1, n_valid * cpl).astype(np.float32)
    yw, nw, bl = ref.fastconv(x, h, blk, ref_flags, 1)
    assert bl == B
    y, n = fc.apply(x, True)                                     # host pointers (the reference's entry)
    assert n == nw, (flags, n, nw)
    lim = (yw.max() - yw.min()) / 1e5                            # tests/test_pffastconv.c:685
    assert np.abs(y - yw).max() <= lim, (flags, taps, blk)
    xd = torch.from_numpy(x).cuda()
    yd = torch.full_like(xd, 7.0)
    y2, n2 = fc.apply(xd, True, out=yd)
    assert n2 == nw and np.abs(y2.cpu().numpy() - yw).max() <= lim
    assert bool((yd[n2 * cpl:] == 7.0).all())                    # nothing beyond the produced samples
    fc.close()


# ------------------------------------------------------------------ per-stream scratch: ten streams on one beyond-LDS setup
@pytest.mark.parametrize("dt,tr,N", [("f32", pa.COMPLEX, 1 << 16), ("f64", pa.REAL, 1 << 17), ("f32", pa.COMPLEX, 36864)])
def test_ten_streams_round_robin_on_one_big_setup(ref, dt, tr, N):
    """A setup beyond LDS keeps its two work buffers per stream and evicts the stream that used it longest ago once more than eight have
    (one entry, not the map).  Ten streams, three rounds, every stream its own input and a batch that differs from round to
    round (the buffers also grow): every result equals the single-stream one bit for bit, and that one meets the reference."""
    from conftest import relerr, tol_for
    dtype = np.float32 if dt == "f32" else np.float64
    tdt = torch.float32 if dt == "f32" else torch.float64
    s = pa.Setup(N, tr, dtype)
    rs = ref.setup(N, tr, dtype)
    streams = [torch.cuda.Stream() for _ in range(10)]
    xs = [_uniform((7, s.vec_scalars), 100 + i, tdt) for i in range(10)]
    want = [s.transform_batch(x, None, pa.FORWARD, i % 2 == 0).clone() for i, x in enumerate(xs)]
    torch.cuda.synchronize()
    for i in (0, 9):
        e = relerr(want[i][:2].cpu().numpy(), rs.batch(xs[i][:2].cpu().numpy(), pa.FORWARD, i % 2 == 0))
        assert e <= tol_for(dt, N), (dt, tr, N, i, e)
    for rnd in range(3):
        k = (3, 7, 5)[rnd]
        outs = []
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs.append(s.transform_batch(xs[i][:k].contiguous(), None, pa.FORWARD, i % 2 == 0))
        torch.cuda.synchronize()
        for i in range(10):
            assert torch.equal(outs[i], want[i][:k]), (rnd, i)
    s.close(); rs.close()


# ------------------------------------------------------------------ beyond LDS on LONG ragged batches
@pytest.mark.parametrize("dt,tr,N", [("f32", pa.REAL, 1 << 17), ("f64", pa.REAL, 1 << 16), ("f32", pa.REAL, 2 * 36864), ("f32", pa.COMPLEX, 10800),
                                     ("f64", pa.COMPLEX, 291600 // 9), ("f32", pa.REAL, 2 * 291600 // 9), ("f32", pa.COMPLEX, 1 << 15)])
def test_beyond_lds_on_long_ragged_batches(ref, dt, tr, N):
    """Real transforms beyond LDS in both directions and the run-time tile plans on a batch of more than four tiles per resident workgroup
    (the in-order loop with its static first tiles, the counter for the rest, a ragged tail): 512 sampled vectors + the ends against
    oracle/_ref, four direction x layout combinations, ordered == zreorder(unordered) and in place == out of place bit for bit."""
    from conftest import relerr, tol_for
    dtype = np.float32 if dt == "f32" else np.float64
    tdt = torch.float32 if dt == "f32" else torch.float64
    s = pa.Setup(N, tr, dtype)
    rs = ref.setup(N, tr, dtype)
    vb = s.vec_scalars * np.dtype(dtype).itemsize
    B = max(600, (640 << 20) // vb) + 3                           # >= 640 MiB of vectors: > 4 tiles per resident workgroup for every tile size
    x = _uniform((B, s.vec_scalars), 77 + N % 1000, tdt)
    rng = np.random.default_rng(N)
    idx = sorted({0, 1, B // 2, B - 2, B - 1} | set(rng.integers(0, B, 512).tolist()))
    it = torch.tensor(idx, device="cuda")
    xh = x[it].cpu().numpy()
    for d in (pa.FORWARD, pa.BACKWARD):
        yo = s.transform_batch(x, None, d, True)
        e = relerr(yo[it].cpu().numpy(), rs.batch(xh, d, True))
        assert e <= tol_for(dt, N), (dt, tr, N, d, "ordered", e)
        yu = s.transform_batch(x, None, d, False)
        e = relerr(yu[it].cpu().numpy(), rs.batch(xh, d, False))
        assert e <= tol_for(dt, N), (dt, tr, N, d, "unordered", e)
        if d == pa.FORWARD:
            assert torch.equal(s.zreorder_batch(yu, None, pa.FORWARD), yo), "ordered != zreorder(unordered)"
        del yu
        xi = x.clone()
        assert torch.equal(s.transform_batch(xi, xi, d, True), yo), "in place != out of place"
        del xi, yo
    s.close(); rs.close()


# ------------------------------------------------------------------ two host threads x two setups, each selecting device 0 itself
def test_two_threads_two_setups_select_the_device_themselves(ref):
    """A setup binds to the device that is current at its first transform (include/pffft_hip.h); the occupancy / LDS opt-in tables are per
    (device, kernel).  Two host threads call hipSetDevice(0) themselves, create two setups each (LDS-resident and beyond LDS) and run them
    concurrently on their own streams: values against the reference, no error counted."""
    from conftest import relerr
    L = pa.lib()
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    errs, before = [], pa.error_count()

    def work(tid):
        try:
            assert hip.hipSetDevice(0) == 0
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for N, tr in ((1024, pa.COMPLEX), (1 << 16, pa.REAL)):
                    s = pa.Setup(N, tr)
                    rs = ref.setup(N, tr, np.float32)
                    x = _uniform((33, s.vec_scalars), 900 + tid)
                    for rep in range(4):
                        y = s.transform_batch(x, None, pa.FORWARD, bool(rep & 1))
                        st.synchronize()
                        e = relerr(y[:3].cpu().numpy(), rs.batch(x[:3].cpu().numpy(), pa.FORWARD, bool(rep & 1)))
                        assert e <= 1e-5, (tid, N, rep, e)
                    s.close(); rs.close()
        except Exception as ex:   # noqa: BLE001
            errs.append((tid, repr(ex)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    assert pa.error_count() == before
    del L


# ------------------------------------------------------------------ HIP graphs next to direct calls
def test_graph_replay_survives_larger_direct_calls_and_more_streams(ref):
    """A captured launch freezes the per-stream scratch pointers and the work-counter address it was recorded with.  Scratch a graph has
    recorded is pinned: a later direct call with a LARGER batch on the same stream retires the outgrown buffers instead of freeing them, and
    the eviction of idle streams skips the recorded entry; captured launches draw their counters from a region of their own, so a replay on
    one stream while direct launches of the same setup run on another shares no counter with them.  The replay keeps writing the values of
    the direct call."""
    st = torch.cuda.Stream()
    N = 1 << 16
    s = pa.Setup(N, pa.COMPLEX)
    with torch.cuda.stream(st):
        x = _uniform((40, 2 * N), 5)
        y = torch.empty_like(x)
        s.transform_batch(x, y, pa.FORWARD, True)                # warm-up: tables, this stream's scratch
        st.synchronize()
        want = y.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            s.transform_batch(x, y, pa.FORWARD, True)
        # (a) a larger direct batch on the same stream: the scratch grows
        xb = _uniform((96, 2 * N), 6)
        yb = s.transform_batch(xb, None, pa.FORWARD, True)
        st.synchronize()
        y.zero_(); g.replay(); st.synchronize()
        assert torch.equal(y, want), "replay after the scratch grew"
    # (b) nine more streams use the setup: the recorded entry is never the eviction victim
    others = [torch.cuda.Stream() for _ in range(9)]
    for o in others:
        with torch.cuda.stream(o):
            s.transform_batch(x[:3].contiguous(), None, pa.FORWARD, True)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        y.zero_(); g.replay(); st.synchronize()
        assert torch.equal(y, want), "replay after nine other streams"
    # (c) replays on `st` while direct in-order launches of the same setup run on another stream: > one ring of counter slots apart is
    #     not needed - the captured launch's counters are outside the ring
    o = others[0]
    big = _uniform((300, 2 * N), 8)
    with torch.cuda.stream(o):
        wbig = s.transform_batch(big, None, pa.FORWARD, True).clone()
    torch.cuda.synchronize()
    for _ in range(5):
        with torch.cuda.stream(st):
            y.zero_(); g.replay()
        with torch.cuda.stream(o):
            ob = s.transform_batch(big, None, pa.FORWARD, True)
        torch.cuda.synchronize()
        assert torch.equal(y, want) and torch.equal(ob, wbig)
    rs = ref.setup(N, pa.COMPLEX, np.float32)
    from conftest import relerr
    assert relerr(want[:2].cpu().numpy(), rs.batch(x[:2].cpu().numpy(), pa.FORWARD, True)) <= 1e-5
    del yb
    rs.close(); s.close()


# ------------------------------------------------------------------ the headline size: short launches
def test_c1024_short_launch_kernel_is_bit_identical(ref):
    """N = 1024 complex float: launches of up to a few resident sets run ONE transform per wavefront in dispatch order
    (fft_c1024_f32_once_kernel), longer ones the persistent in-order loop.  Same part A / part B: the spectrum of a vector does not depend
    on the batch it travelled in - every batch size around the switch, four direction x layout combinations, against the long batch bit
    for bit and against the reference."""
    from conftest import relerr
    s = pa.Setup(1024, pa.COMPLEX)
    rs = ref.setup(1024, pa.COMPLEX, np.float32)
    big = 1 << 17
    x = _uniform((big, 2048), 12)
    for d in (pa.FORWARD, pa.BACKWARD):
        for o in (True, False):
            full = s.transform_batch(x, None, d, o)
            assert relerr(full[:64].cpu().numpy(), rs.batch(x[:64].cpu().numpy(), d, o)) <= 1e-5
            for k in (1, 3, 4, 5, 63, 1024, 4096, 4097, 16384, 16385, 65536):
                part = s.transform_batch(x[:k].contiguous(), None, d, o)
                assert torch.equal(part, full[:k]), (d, o, k)
            xi = x[:5000].clone()
            assert torch.equal(s.transform_batch(xi, xi, d, o), full[:5000]), "in place"
    s.close(); rs.close()


# ------------------------------------------------------------------ pffastconv: first use from two streams
def test_fastconv_first_use_from_two_streams(ref):
    """The folded coefficient table of the few-block kernel is built on first use: complete before its pointer is published (null stream +
    synchronisation), whatever stream the first call came from - a second stream's first call must not read a half-built table."""
    rng = np.random.default_rng(3)
    h = rng.uniform(-1, 1, 4096).astype(np.float32)
    x = rng.uniform(-1, 1, 1 << 20).astype(np.float32)
    yw, nw, _ = ref.fastconv(x, h, 0, 0, 1)
    xd = torch.from_numpy(x).cuda()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for rep in range(3):
        fc = pa.FastConv(h, 0, 0)                                # a fresh setup: first use every time
        with torch.cuda.stream(sa):
            ya, na = fc.apply(xd, True)
        with torch.cuda.stream(sb):
            yb, nb = fc.apply(xd, True)
        torch.cuda.synchronize()
        lim = (yw.max() - yw.min()) / 1e5
        assert na == nw and nb == nw
        assert np.abs(ya.cpu().numpy() - yw).max() <= lim and np.abs(yb.cpu().numpy() - yw).max() <= lim, rep
        fc.close()


# ------------------------------------------------------------------ one host thread, several parts / devices (C-level sharding)
@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_transform_batch_multi_from_one_thread(ref, dt):
    """pffft[d]_hip_transform_batch_multi: batch shards driven from ONE host thread - per part hipSetDevice + the batched entry on its own
    setup and stream.  A 1-GPU box has one device: three parts on device 0 (three setups, three streams, ragged shard sizes) must equal the
    one-call result bit for bit and meet the reference; a device that does not exist is an error, not a fault; the current device is
    restored."""
    import ctypes as C
    from conftest import relerr
    L = pa.lib()
    dtype = np.float64 if dt == "f64" else np.float32
    multi = L.pffftd_hip_transform_batch_multi if dt == "f64" else L.pffft_hip_transform_batch_multi
    multi.restype = C.c_int
    N, B = 1024, 3000
    x = _uniform((B, 2 * N), 21, torch.float64 if dt == "f64" else torch.float32)
    whole = pa.Setup(N, pa.COMPLEX, dtype)
    want = whole.transform_batch(x, None, pa.FORWARD, False)
    torch.cuda.synchronize()
    cuts = [0, 1000, 1001, B]
    setups = [pa.Setup(N, pa.COMPLEX, dtype) for _ in range(3)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    y = torch.zeros_like(x)
    devs = (C.c_int * 3)(0, 0, 0)
    hs = (C.c_void_p * 3)(*[s.handle for s in setups])
    ins = (C.c_void_p * 3)(*[x[cuts[i]:].data_ptr() for i in range(3)])
    outs = (C.c_void_p * 3)(*[y[cuts[i]:].data_ptr() for i in range(3)])
    bs = (C.c_size_t * 3)(*[cuts[i + 1] - cuts[i] for i in range(3)])
    sts = (C.c_void_p * 3)(*[s.cuda_stream for s in streams])
    rc = multi(3, devs, hs, ins, outs, bs, pa.FORWARD, 0, sts)
    assert rc == 0, pa.last_error()
    torch.cuda.synchronize()
    assert torch.equal(y, want)
    rs = ref.setup(N, pa.COMPLEX, dtype)
    assert relerr(y[[0, 1000, B - 1]].cpu().numpy(), rs.batch(x[[0, 1000, B - 1]].cpu().numpy(), pa.FORWARD, False)) <= (1e-13 if dt == "f64" else 1e-5)
    bad = (C.c_int * 1)(97)
    assert multi(1, bad, hs, ins, outs, bs, pa.FORWARD, 0, None) != 0
    assert torch.cuda.current_device() == 0
    whole.transform_batch(x[:4].contiguous(), None, pa.FORWARD, True)     # the library still works on the restored device
    torch.cuda.synchronize()
    for s in setups + [whole]:
        s.close()
    rs.close()
