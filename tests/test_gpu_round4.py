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
@pytest.mark.parametrize("taps,L,nsig", [(4096, (1 << 22) + 4321, 1), (1025, (1 << 22) + 17, 2), (2048, 3 * (1 << 20) + 3, 3),
                                        (3001, 1 << 22, 2), (8192, (1 << 22) + 5, 1), (5000, 2500001, 2)])
@pytest.mark.parametrize("flush", [1, 0])
def test_fastconv_split_kernel(ref, taps, L, nsig, flush):
    """fastconv_split_kernel - default for 16384-sample internal blocks (filters beyond 1024 taps on calls with many blocks) and
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
        for var in (0, 116, 97):
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


@pytest.mark.parametrize("taps,L", [(4096, 1 << 20), (2048, 1 << 19), (1500, 300001), (3000, 700003)])
def test_fastconv_few_blocks_512_thread_configurations(ref, taps, L):
    """Calls with few blocks (the stated C4 call: 2^20 samples, 4096 taps) run one workgroup per reference-sized block; round 4:
    512 / 256 threads per 8192- / 4096-sample block (FirCfg::C4096m / C2048m, eight points per thread), the first block's samples
    requested before the tables.  Variant 114 = the 16-points-per-thread configurations."""
    rng = np.random.default_rng(taps)
    x = rng.uniform(-1, 1, L).astype(np.float32)
    h = rng.uniform(-1, 1, taps).astype(np.float32)
    fc = pa.FastConv(h, 0, 0)
    xd = torch.from_numpy(x).cuda()
    try:
        for flush in (1, 0):
            yw, nw, _ = ref.fastconv(x, h, 0, 0, flush)
            for var in (0, 114):
                pa.set_variant(var)
                yd = torch.full_like(xd, 7.0)
                y, n = fc.apply(xd, bool(flush), out=yd)
                assert n == nw, (var, flush)
                assert np.abs(y.cpu().numpy() - yw).max() <= (yw.max() - yw.min()) / 1e5, (var, flush)
                assert bool((yd[n:] == 7.0).all())
    finally:
        pa.set_variant(0)
    fc.close()
