"""CPU tests (-m "not gpu"): pin the numpy restatement oracle/pffft_oracle.py against
 (1) the committed fixtures generated from the real reference (tests/golden/make_golden.py),
 (2) the real reference itself (oracle/_ref) when present,
 (3) the reference's own generative checks: analytic single-tone spectra (tests/test_pffft.c:109-247),
     accepted-size set (tests/test_fft_factors.c:36-61), next/is_power_of_two tables
     (tests/test_pffft.c:280-330), naive FIR (tests/test_pffastconv.c:175-213)."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES, gkey, relerr, tol_for
from oracle import pffft_oracle as po


def _dt(dt):
    return np.float32 if dt == "f32" else np.float64


@pytest.mark.parametrize("dt,tr,N", GOLDEN_CASES)
def test_oracle_matches_golden(golden, dt, tr, N):
    k, dtype, tol = gkey(dt, tr, N), _dt(dt), tol_for(dt, N)
    x = golden[k + "_x"]
    fo = po.transform(x, N, tr, po.FORWARD, True, dtype)
    fu = po.transform(x, N, tr, po.FORWARD, False, dtype)
    assert relerr(fo, golden[k + "_fwd_ordered"]) <= tol
    assert relerr(fu, golden[k + "_fwd_unordered"]) <= tol
    bo = po.transform(golden[k + "_fwd_ordered"], N, tr, po.BACKWARD, True, dtype)
    bu = po.transform(golden[k + "_fwd_unordered"], N, tr, po.BACKWARD, False, dtype)
    assert relerr(bo, golden[k + "_bwd_ordered"]) <= tol
    assert relerr(bu, golden[k + "_bwd_unordered"]) <= tol
    # unscaled round trip: BACKWARD(FORWARD(x)) = N x  (include/pffft/pffft.h:134)
    assert relerr(bu / N, x) <= 10 * tol


@pytest.mark.parametrize("dt,tr,N", [c for c in GOLDEN_CASES if c[2] <= 1024])
def test_layout_and_zconvolve_match_golden(golden, dt, tr, N):
    k, dtype = gkey(dt, tr, N), _dt(dt)
    perm = po.internal_index_table(N, tr)
    assert np.array_equal(perm, golden[k + "_perm"])  # closed form == pffft_zreorder's permutation
    fu = golden[k + "_fwd_unordered"]
    assert np.array_equal(po.zreorder(fu, N, tr, po.FORWARD), golden[k + "_fwd_ordered"]) or \
        relerr(po.zreorder(fu, N, tr, po.FORWARD), golden[k + "_fwd_ordered"]) <= 1e-6
    assert np.array_equal(po.zreorder(po.zreorder(fu, N, tr, po.FORWARD), N, tr, po.BACKWARD), fu)
    for acc, name in ((True, "_zc_accumulate"), (False, "_zc_no_accu")):
        got = po.zconvolve(fu, golden[k + "_zc_b"], golden[k + "_zc_acc0"], 0.25, tr, acc, dtype)
        assert relerr(got, golden[k + name]) <= (1e-6 if dt == "f32" else 1e-14)


@pytest.mark.parametrize("name", ["ramp", "rand", "cplx2", "cplx1", "corr"])
@pytest.mark.parametrize("flush", [0, 1])
def test_fastconv_oracle_matches_golden(golden, name, flush):
    k = f"fc_{name}_flush{flush}"
    L, taps, blk, flags, fl, n, bl = [int(v) for v in golden[k + "_meta"]]
    s = po.fastconv_setup(golden[k + "_h"], blk, flags)
    assert s["blockLen"] == bl
    y, produced = po.fastconv_apply(s, golden[k + "_x"], bool(flush))
    assert produced == n
    want = golden[k + "_y"]
    lim = (want.max() - want.min()) / 1e5 if want.size else 0  # tests/test_pffastconv.c:685
    assert np.abs(y - want).max() <= max(lim, 1e-30) if want.size else True


def test_fastconv_against_naive_fir(golden):
    # the reference test's ground truth (slow_conv_R): only for the real, non-correlation cases
    for name in ("ramp", "rand"):
        k = f"fc_{name}_flush1"
        x, h, y = golden[k + "_x"], golden[k + "_h"], golden[k + "_y"]
        naive = po.slow_conv(x, h)
        assert y.size == naive.size
        lim = (naive.max() - naive.min()) / 1e5
        assert np.abs(y - naive).max() <= lim


def test_oracle_against_real_reference(ref):
    """Fresh seeds, more sizes (the reference's validation list, benchmarks/bench_pffft.c:445)."""
    rng = np.random.default_rng(7)
    for dt in ("f32", "f64"):
        dtype = _dt(dt)
        for tr, sizes in ((po.COMPLEX, [16, 32, 64, 96, 128, 160, 192, 256, 288, 384, 480, 512, 576, 640, 800, 864,
                                        1024, 2048, 2592, 4000, 4096]),
                          (po.REAL, [32, 64, 96, 128, 160, 192, 256, 288, 384, 480, 512, 576, 640, 800, 864, 1024,
                                     2048, 2592 * 4, 4000, 4096, 12000])):
            for N in sizes:
                if not po.new_setup_ok(N, tr):
                    continue
                s = ref.setup(N, tr, dtype)
                x = rng.uniform(-1, 1, s.nfloats).astype(dtype)
                tol = tol_for(dt, N)
                for ordered in (False, True):
                    want = (s.transform_ordered if ordered else s.transform_unordered)(x, 0)
                    assert relerr(po.transform(x, N, tr, 0, ordered, dtype), want) <= tol, (dt, tr, N, ordered)
                    wb = (s.transform_ordered if ordered else s.transform_unordered)(want, 1)
                    assert relerr(po.transform(want, N, tr, 1, ordered, dtype), wb) <= tol, (dt, tr, N, ordered)
                s.close()


def test_size_helpers_match_reference(ref):
    """tests/test_fft_factors.c:36-61 restated: validity == setup-non-NULL, and both agree with the reference."""
    for api in (ref.f32, ref.f64):
        for tr in (po.REAL, po.COMPLEX):
            nmin = api.min_fft_size(tr)
            assert nmin == po.min_fft_size(tr)
            for N in range(nmin // 2, 12 * nmin + 1, nmin // 2):
                assert bool(api.is_valid_size(N, tr)) == po.is_valid_size(N, tr), (N, tr)
                h = api.new_setup(N, tr)
                assert bool(h) == po.new_setup_ok(N, tr), (N, tr)
                if h:
                    api.destroy_setup(h)
            for N in (1, 17, 100, 1000, 5000):
                for higher in (0, 1):
                    assert api.nearest_transform_size(N, tr, higher) == po.nearest_transform_size(N, tr, bool(higher))
        for N in list(range(0, 70)) + [255, 256, 257, 1 << 20, (1 << 20) + 1]:
            assert api.next_power_of_two(N) == po.next_power_of_two(N), N
            assert bool(api.is_power_of_two(N)) == po.is_power_of_two(N), N


def test_power_of_two_tables():
    # tests/test_pffft.c:280-330
    assert [po.next_power_of_two(n) for n in (1, 2, 3, 4, 5, 7, 8, 9, 17, 1023, 1024, 1025)] == \
        [1, 2, 4, 4, 8, 8, 8, 16, 32, 1024, 1024, 2048]
    assert [po.is_power_of_two(n) for n in (0, 1, 2, 3, 4, 6, 8, 1024, 1025)] == \
        [False, True, True, False, True, False, True, True, False]


@pytest.mark.parametrize("N", [32, 64, 256, 1024, 4096])
@pytest.mark.parametrize("cplx", [0, 1])
def test_single_tone_spectrum(N, cplx):
    """tests/test_pffft.c:109-247: a tone at bin k gives one carrier bin, everything else >= 140 dB
    (float) below; magnitude error <= 1e-6 relative; round trip sum-square error <= N*1e-7."""
    for kk in (0, N // 16, 3 * N // 16):
        amp, phi0 = 1.1, np.pi / 8
        n = np.arange(N)
        if cplx:
            z = amp * np.exp(1j * (2 * np.pi * kk * n / N + phi0))
            x = np.empty(2 * N, np.float32); x[0::2], x[1::2] = z.real, z.imag
            X = po.transform(x, N, po.COMPLEX, po.FORWARD, True)
            P = X[0::2].astype(np.float64) ** 2 + X[1::2].astype(np.float64) ** 2
            carrier = kk
            expected = (amp * N) ** 2
        else:
            x = (amp * np.cos(2 * np.pi * kk * n / N + phi0)).astype(np.float32)
            X = po.transform(x, N, po.REAL, po.FORWARD, True)
            P = X[0::2].astype(np.float64) ** 2 + X[1::2].astype(np.float64) ** 2
            if kk == 0:
                P[0] = float(X[0]) ** 2  # DC only; X[1] is the Nyquist bin
            carrier = kk
            expected = (amp * np.cos(phi0) * N) ** 2 if kk == 0 else (amp * N / 2) ** 2
        others = np.delete(P, carrier)
        assert P[carrier] > 0
        assert 10 * np.log10(P[carrier] / max(others.max(), 1e-300)) >= 140 - 30 * cplx * 0  # dynamic range
        assert abs(np.sqrt(P[carrier]) / np.sqrt(expected) - 1) <= 1e-5
        back = po.transform(X, N, po.COMPLEX if cplx else po.REAL, po.BACKWARD, True)
        assert np.sum((back / N - x) ** 2) <= N * 1e-7


# ------------------------------------------------------------------ BASELINE configs[0] (C1): the PFFFT_USE_SIMD=OFF builds
def _c1_inputs():
    """N=64 real float forward, x[i] = hash32(seed=1, i)/2^31 - 1 in [-1, 1) and the single-tone
    case of tests/test_pffft.c:126-140 (cosine at bin k, amplitude 1, phase 0.3 rad)."""
    i = np.arange(64, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(1) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF); h ^= h >> np.uint64(13)
    x_hash = (h.astype(np.float64) / 2.0 ** 31 - 1.0).astype(np.float32)
    k = 5
    x_tone = np.cos(2 * np.pi * k * np.arange(64) / 64 + 0.3).astype(np.float32)
    return {"hash": x_hash, "tone": x_tone}


def test_c1_4xscalar_build_is_bit_identical_to_sse():
    """The survey's finding 3 ("every SIMD_SZ == 4 build gives bit-identical results") as a committed test for C1:
    -DPFFFT_SIMD_DISABLE=1 -DPFFFT_SCALVEC_ENABLED=1 (CMakeLists.txt:19-20,172-181) vs the SSE object, N=64 real forward,
    ordered and unordered, plus the inverse; and both against the numpy restatement."""
    import os
    from conftest import missing_checker
    from oracle import ref as oref
    if not oref.available():
        oref.build()
    for so in (oref.REF_SO, oref.REF_4XSCALAR_SO):
        if not os.path.exists(so):
            missing_checker(so)
    sse, s4 = oref.get(), oref.Reference(oref.REF_4XSCALAR_SO)
    assert s4.f32.simd_arch().decode() == "4xScalar" and s4.f32.simd_size() == 4
    assert sse.f32.simd_size() == 4 and sse.f32.simd_arch().decode() != "4xScalar"
    a, b = sse.setup(64, 0), s4.setup(64, 0)
    for name, x in _c1_inputs().items():
        fu_a, fu_b = a.transform_unordered(x, 0), b.transform_unordered(x, 0)
        fo_a, fo_b = a.transform_ordered(x, 0), b.transform_ordered(x, 0)
        assert np.array_equal(fu_a, fu_b) and np.array_equal(fo_a, fo_b), name
        assert np.array_equal(a.transform_unordered(fu_a, 1), b.transform_unordered(fu_b, 1)), name
        assert np.array_equal(a.zreorder(fu_a, 0), fo_b), name
        assert relerr(po.transform(x, 64, po.REAL, po.FORWARD, True, np.float32), fo_b) <= 1e-6, name
        assert relerr(po.transform(x, 64, po.REAL, po.FORWARD, False, np.float32), fu_b) <= 1e-6, name
    # the analytic answer of the tone: bin 5 = (N/2) e^{i 0.3}
    fo = b.transform_ordered(_c1_inputs()["tone"], 0)
    assert abs(fo[10] - 32 * np.cos(0.3)) <= 1e-4 and abs(fo[11] - 32 * np.sin(0.3)) <= 1e-4
    a.close(); b.close()


def test_c1_scalar_build_pins_the_fftpack_order_layout():
    """SCALAR_VECT=OFF as well (SIMD_SZ == 1): simd_size 1, min sizes 2 / 1, ordered output equal (to rounding) to the
    SIMD_SZ == 4 builds', and the UNORDERED output in FFTPACK's half-complex order r0, r1, i1, ..., r_{N/2}
    (src/pffft_priv_impl.h:1712-1766: the scalar path's transform IS the ordered transform up to the canonical pack) —
    the layout SURVEY.md §8(c) warns not to use for unordered / convolution parity."""
    import os
    from conftest import missing_checker
    from oracle import ref as oref
    if not os.path.exists(oref.REF_SCALAR_SO):
        oref.build()
    if not os.path.exists(oref.REF_SCALAR_SO):
        missing_checker(oref.REF_SCALAR_SO)
    sc, sse = oref.Reference(oref.REF_SCALAR_SO), oref.get()
    assert sc.f32.simd_size() == 1 and sc.f32.min_fft_size(0) == 2 and sc.f32.min_fft_size(1) == 1
    a, b = sc.setup(64, 0), sse.setup(64, 0)
    for name, x in _c1_inputs().items():
        fo = a.transform_ordered(x, 0)
        assert relerr(fo, b.transform_ordered(x, 0)) <= 1e-6, name
        z = np.fft.rfft(x.astype(np.float64))
        want_ordered = np.empty(64); want_ordered[0] = z[0].real; want_ordered[1] = z[32].real
        want_ordered[2::2] = z[1:32].real; want_ordered[3::2] = z[1:32].imag
        assert relerr(fo, want_ordered) <= 1e-6, name
        fu = a.transform_unordered(x, 0)
        want_fftpack = np.empty(64); want_fftpack[0] = z[0].real; want_fftpack[63] = z[32].real
        want_fftpack[1:63:2] = z[1:32].real; want_fftpack[2:63:2] = z[1:32].imag
        assert relerr(fu, want_fftpack) <= 1e-6, name
        assert not np.allclose(fu, b.transform_unordered(x, 0))   # NOT the SIMD_SZ == 4 internal layout
    a.close(); b.close()
