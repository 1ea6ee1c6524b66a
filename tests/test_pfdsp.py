"""GPU parity tests (-m gpu) for SURVEY.md §8 row f-4: the PFDSP mixers on MI355X (libpfdsp_hip.so, reached
through the reference's own C entry points, include/pfdsp_hip.h) against
  * the float64 closed form every reference algorithm approximates (oracle/pfdsp_oracle.exact) — bar 1e-6
    absolute on |x| <= sqrt(2), at ANY stream position (the reference cannot be the yardstick there: its
    float phase accumulators drift, see tests/test_pfdsp_oracle.py);
  * the REAL reference compiled from src/pf_mixer.cpp (oracle/_ref/libpfdsp_ref.so) on the same inputs —
    bar 1e-5 relative (BASELINE.json north_star, float) at the block sizes where the reference's own drift
    is below it, and the reference's measured drift bound DRIFT(n) for longer streams;
  * each algorithm's chaining contract: returned phase / advanced state struct feed the next call.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import pfdsp_oracle as mo  # noqa: E402
from oracle import pfdsp_ref  # noqa: E402
from pffft_amd import pfdsp  # noqa: E402

RATE, PH0 = 0.0137, 0.4
TIGHT = 1e-6


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device: the product has no CPU fallback")
    torch.cuda.set_device(0)
    pfdsp.lib()


@pytest.fixture(scope="module")
def R():
    if not pfdsp_ref.available():
        __import__("conftest").missing_checker("oracle/_ref/libpfdsp_ref.so")
    return pfdsp_ref.get()


def _x(n, seed=7):
    rng = np.random.default_rng(seed)
    return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)


def _maxerr(a, b):
    return float(np.abs(np.asarray(a, np.complex128) - np.asarray(b, np.complex128)).max())


def _osc(m):
    """(lane phasors, block angle) of a recursive-oscillator mixer as its state struct holds them NOW"""
    k = 8 if m.algo == "recursive_osc" else 4
    S = np.array(m.data.u_cos[:k], np.float64) + 1j * np.array(m.data.v_sin[:k], np.float64)
    return S, mo.osc_step_angle(m.conf.k1, m.conf.k2)


def _want(algo, x, rate, ph0, first_sample=0, osc=None):
    """closed form of one call on a FRESH mixer of this algorithm (sample offset as the algorithm defines it);
    recursive oscillators: from the lane phasors `osc` captured before the call"""
    i = np.arange(x.size) + first_sample
    if algo.startswith("recursive"):
        S, th = osc
        return x.astype(np.complex128) * S[i % S.size] * np.exp(1j * th * (i // S.size))
    return mo.exact(x, mo.increment(rate), ph0, first=first_sample + (1 if algo == "addfast" else 0))


@pytest.mark.parametrize("algo", pfdsp.ALGOS)
@pytest.mark.parametrize("rate", [0.0137, -0.21, 0.499])
def test_against_closed_form_and_reference(R, algo, rate):
    for n in (256, 4096):
        x = _x(n, n)
        m = pfdsp.Mixer(algo, rate, PH0)
        osc = _osc(m) if algo.startswith("recursive") else None
        y = m(x)                                  # host pointers: staged through the device
        want = _want(algo, x, rate, PH0, osc=osc)
        assert _maxerr(y, want) <= TIGHT, (algo, n)
        if algo != "table":                       # reference's table variant: quadrant resolution (src/pf_mixer.cpp:202)
            mr = pfdsp.Mixer(algo, rate, PH0, abi=R)
            yr = mr(x)
            assert _maxerr(y, yr) <= mo.DRIFT(n), (algo, n)
            if n == 256 and abs(rate) < 0.3:
                assert _maxerr(y, yr) <= 1e-5 * np.abs(yr).max(), (algo, n)   # north_star's float bar
            # the state the next call starts from
            if algo in ("math", "addfast", "unroll"):
                d = abs(m.phase - mr.phase)
                # A accumulates the returned phase sample by sample (DRIFT); C and D form it as the float product
                # n*inc and wrap it with a float 2*pi, one rounded subtraction per turn (src/pf_mixer.cpp:280-283)
                bar = mo.DRIFT(n) if algo == "math" else mo.RETURN_BAR(n, mo.increment(rate))
                assert min(d, abs(d - 2 * np.pi)) <= bar, (algo, m.phase, mr.phase, bar)
            if algo == "limited_unroll":
                assert abs(m.data.complex_phase.i - mr.data.complex_phase.i) <= mo.DRIFT(n)
                assert abs(m.data.complex_phase.q - mr.data.complex_phase.q) <= mo.DRIFT(n)
            if algo.endswith("_sse") and algo.startswith("limited"):
                assert np.abs(np.array(m.data.phase_state_i[:]) - np.array(mr.data.phase_state_i[:])).max() <= mo.DRIFT(n)
                assert np.abs(np.array(m.data.phase_state_q[:]) - np.array(mr.data.phase_state_q[:])).max() <= mo.DRIFT(n)
            if algo.startswith("recursive"):
                k = 8 if algo == "recursive_osc" else 4
                assert np.abs(np.array(m.data.u_cos[:k]) - np.array(mr.data.u_cos[:k])).max() <= mo.DRIFT(n)
                assert np.abs(np.array(m.data.v_sin[:k]) - np.array(mr.data.v_sin[:k])).max() <= mo.DRIFT(n)
            mr.close()
        m.close()


@pytest.mark.parametrize("algo", pfdsp.ALGOS)
def test_chained_calls_continue_the_stream(algo):
    """three calls of unequal length == one call over the concatenation (phase / state carried by the caller)"""
    x = _x(3000, 3)
    cuts = (0, 1000, 1000 + 136, 3000)           # multiples of 8: every algorithm's block size
    m1, m2 = pfdsp.Mixer(algo, RATE, PH0), pfdsp.Mixer(algo, RATE, PH0)
    osc = _osc(m1) if algo.startswith("recursive") else None
    whole = m1(x)
    parts = np.concatenate([m2(np.ascontiguousarray(x[a:b])) for a, b in zip(cuts[:-1], cuts[1:])])
    assert _maxerr(whole, parts) <= 2 * TIGHT
    assert _maxerr(whole, _want(algo, x, RATE, PH0, osc=osc)) <= TIGHT
    m1.close(); m2.close()


@pytest.mark.parametrize("algo", ["addfast", "unroll", "limited_unroll", "recursive_osc"])
def test_inplace_entries_and_device_pointers(algo):
    x = _x(2048, 5)
    m = pfdsp.Mixer(algo, RATE, PH0)
    want = m(x)
    m2 = pfdsp.Mixer(algo, RATE, PH0)
    buf = x.copy()
    m2(buf, inplace=True)                         # *_inp_c entry, host pointer
    assert np.array_equal(buf, want)
    m3 = pfdsp.Mixer(algo, RATE, PH0)
    xd = torch.from_numpy(x).cuda()
    yd = m3(xd)                                   # device pointers used in place, no staging
    assert np.array_equal(yd.cpu().numpy(), want)
    m4 = pfdsp.Mixer(algo, RATE, PH0)
    m4(xd, inplace=True)
    assert np.array_equal(xd.cpu().numpy(), want)
    for mm in (m, m2, m3, m4):
        mm.close()


def test_gen_recursive_osc():
    m = pfdsp.Mixer("recursive_osc", RATE, PH0)
    osc = _osc(m)
    g = np.empty(1024, np.complex64)
    m.generate(g)
    want = _want("recursive_osc", np.ones(1024, np.complex64), RATE, PH0, osc=osc)
    assert _maxerr(g, want) <= TIGHT
    g2 = np.empty(512, np.complex64)
    m.generate(g2)                                # continues where the first call stopped
    want2 = _want("recursive_osc", np.ones(512, np.complex64), RATE, PH0, first_sample=1024, osc=osc)
    assert _maxerr(g2, want2) <= 2 * TIGHT


def test_update_rate_keeps_phase(R):
    """shift_recursive_osc_update_rate mid-stream (src/pf_mixer.cpp:898-921): lane 0 keeps its phasor"""
    x = _x(512, 9)
    out = []
    for abi in (None, R):
        m = pfdsp.Mixer("recursive_osc", RATE, PH0, abi=abi)
        a = m(x)
        m.L.shift_recursive_osc_update_rate(0.05, C.byref(m.conf), C.byref(m.data))
        b = m(x)
        out.append(np.concatenate([a, b]))
    assert _maxerr(out[0], out[1]) <= mo.DRIFT(1024)


def test_unaligned_and_odd_lengths():
    """device pointers that are only 8-byte aligned and odd lengths take the float2 kernel"""
    base = torch.from_numpy(_x(4099, 13)).cuda()
    for off, n in ((1, 4097), (0, 4095), (1, 2), (3, 1)):
        x = base[off:off + n]
        y = pfdsp.shift_device(x, RATE, PH0)
        want = mo.exact(x.cpu().numpy(), 2 * np.pi * RATE, PH0)
        assert _maxerr(y.cpu().numpy(), want) <= TIGHT, (off, n)


def test_long_stream_phase_is_exact():
    """2^26 samples: the phase of the LAST samples is still exact (double reduction), where a float accumulator
    would be off by radians.  Checked on the tail and on a strided sample of the whole stream."""
    n = 1 << 26
    rate = 0.123456789
    x = torch.ones(n, dtype=torch.complex64, device="cuda")
    y = pfdsp.shift_device(x, rate, PH0)
    idx = np.concatenate([np.arange(0, n, 65521), np.arange(n - 4096, n)])
    got = y[torch.from_numpy(idx).cuda()].cpu().numpy()
    t = (rate * idx.astype(np.float64)) % 1.0
    want = np.exp(1j * (2 * np.pi * t + PH0))
    assert _maxerr(got, want) <= TIGHT
    osc = pfdsp.shift_device(None, rate, PH0, out=torch.empty_like(x))     # in == NULL: the oscillator itself
    assert torch.equal(osc, y)


@pytest.mark.parametrize("algo", ["limited_unroll_A_sse", "recursive_osc", "addfast"])
def test_long_stream_lane_algorithms(algo):
    """2^24 + 1000 samples on a device pointer, in place: the streaming kernel (persistent workgroups, in-order chunks)
    for the full 8 KiB chunks + the tail kernel, 4-lane / 8-lane / 1-lane states; checked on a strided sample + the tail,
    and the state must continue into a second call."""
    n = (1 << 24) + 1000
    rng = np.random.default_rng(21)
    xh = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    x = torch.from_numpy(xh).cuda()
    m = pfdsp.Mixer(algo, RATE, PH0)
    osc = _osc(m) if algo.startswith("recursive") else None
    m(x, inplace=True)
    idx = np.concatenate([np.arange(0, n, 4099), np.arange(n - 3000, n)])
    got = x[torch.from_numpy(idx).cuda()].cpu().numpy()
    if algo.startswith("recursive"):
        S, th = osc
        ph = th * (idx // S.size).astype(np.float64)
        want = xh[idx].astype(np.complex128) * S[idx % S.size] * np.exp(1j * ph)
    else:
        inc = float(mo.increment(RATE))
        first = 1 if algo == "addfast" else 0
        want = xh[idx].astype(np.complex128) * np.exp(1j * (PH0 + ((idx + first).astype(np.float64) * inc) % (2 * np.pi)))
    assert _maxerr(got, want) <= 2 * TIGHT
    # second call continues: sample k of call 2 is sample n + k of the stream
    y2 = m(torch.from_numpy(xh[:1024].copy()).cuda()).cpu().numpy()
    k = np.arange(1024) + n
    if algo.startswith("recursive"):
        want2 = xh[:1024].astype(np.complex128) * S[k % S.size] * np.exp(1j * th * (k // S.size).astype(np.float64))
    else:
        want2 = xh[:1024].astype(np.complex128) * np.exp(1j * (PH0 + ((k + first).astype(np.float64) * inc) % (2 * np.pi)))
    assert _maxerr(y2, want2) <= 4 * TIGHT
    m.close()


def test_symbols_resolve_to_hip_library():
    import subprocess
    out = subprocess.run(["nm", "-D", "--undefined-only", pfdsp.lib_path()], capture_output=True, text=True).stdout
    assert "hipLaunchKernel" in out or "hipModuleLaunchKernel" in out or "__hipPushCallConfiguration" in out


# ------------------------------------------------------------------ the mixer fused into the forward FFT
import pffft_amd as pa  # noqa: E402


@pytest.mark.parametrize("N", [1024, 256, 96, 4096])
@pytest.mark.parametrize("ordered", [True, False])
def test_shift_transform_batch(N, ordered):
    """pffft_hip_shift_transform_batch == (mixer kernel, then pffft_hip_transform_batch) == the reference chain
    shift_math_cc -> pffft_transform[_ordered] on the same stream.  N = 1024 takes the fused kernel."""
    from oracle import ref as oref
    batch, rate, ph = 37, -0.0731, 1.234
    s = pa.Setup(N, pa.COMPLEX, np.float32)
    rng = np.random.default_rng(N)
    x = rng.uniform(-1, 1, (batch, 2 * N)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    got = s.shift_transform_batch(xd, rate, ph, ordered=ordered).cpu().numpy()
    # composition on the device
    mixed = pfdsp.shift_device(xd.view(torch.complex64).reshape(-1), rate, ph)
    comp = s.transform_batch(torch.view_as_real(mixed).reshape(batch, 2 * N).contiguous(), None, pa.FORWARD, ordered).cpu().numpy()
    scale = np.abs(comp).max()
    assert np.abs(got - comp).max() <= 2e-6 * scale
    # float64 truth: exact oscillator, numpy FFT (ordered only: canonical layout)
    xc = (x[:, 0::2] + 1j * x[:, 1::2]).astype(np.complex128).reshape(-1)
    g = np.arange(xc.size, dtype=np.float64)
    want = np.fft.fft((xc * np.exp(1j * (ph + 2 * np.pi * ((rate * g) % 1.0)))).reshape(batch, N), axis=1)
    if ordered:
        gc = got[:, 0::2] + 1j * got[:, 1::2]
        assert np.abs(gc - want).max() <= 1e-5 * np.abs(want).max()
    # the reference chain (float mixer, then the reference FFT), when shipped
    if oref.available() and pfdsp_ref.available() and ordered:
        R = pfdsp_ref.get()
        xm = np.empty(batch * N, np.complex64)
        R.shift_math_cc(xc.astype(np.complex64).ctypes.data, xm.ctypes.data, batch * N, rate, ph)
        rs = oref.get().setup(N, pa.COMPLEX, np.float32)
        i = batch - 1                                    # the last vector: furthest from the stream start
        wr = rs.transform_ordered(np.ascontiguousarray(xm[i * N:(i + 1) * N]).view(np.float32), oref.FORWARD)
        rs.close()
        # bar: the reference mixer's own phase drift after batch*N samples (DRIFT, radians — common to the whole
        # vector, so every bin X_k is off by at most DRIFT * |X_k|) + north_star's 1e-5 for the transform itself
        assert np.abs(got[i] - wr).max() <= (mo.DRIFT(batch * N) + 1e-5) * np.abs(wr).max()
    s.close()


def test_shift_transform_far_into_the_stream():
    """fused kernel, 2^16 transforms of 1024: the phase of the last vectors is still exact"""
    N, batch, rate, ph = 1024, 1 << 16, 0.123456789, 0.5
    s = pa.Setup(N, pa.COMPLEX, np.float32)
    x = torch.zeros(batch, 2 * N, device="cuda")
    x[:, 0] = 1.0                                         # an impulse at sample 0 of every vector: flat spectrum = its phasor
    got = s.shift_transform_batch(x, rate, ph, ordered=True)
    sel = [0, 1, 12345, batch - 2, batch - 1]
    g = got[sel].cpu().numpy()
    for row, b in zip(g, sel):
        want = np.exp(1j * (ph + 2 * np.pi * ((rate * (b * N)) % 1.0)))
        assert np.abs((row[0::2] + 1j * row[1::2]) - want).max() <= 2e-6
    s.close()


def test_shift_transform_rejects_other_setups():
    for N, tr, dt in ((1024, pa.REAL, np.float32), (1024, pa.COMPLEX, np.float64)):
        s = pa.Setup(N, tr, dt)
        rc = pa.lib().pffft_hip_shift_transform_batch(s.handle, None, None, 1, 0, 0.1, 0.0, None)
        assert rc != 0 and b"complex single-precision" in pa.lib().pffft_hip_last_error()
        s.close()


def test_against_committed_golden():
    """tests/golden/pfdsp_golden.npz (outputs of the reference's own object code, two chained 256-sample calls per
    algorithm and rate): the HIP path stays within the reference's drift bound of every stored output, with its own
    chained state."""
    import os
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pfdsp_golden.npz")))
    x, ph0 = g["x"], float(g["ph0"][0])
    n = x.size
    for ri, rate in enumerate(g["rates"]):
        for algo in pfdsp.ALGOS:
            if algo == "table":
                continue
            m = pfdsp.Mixer(algo, float(rate), ph0)
            y = np.concatenate([m(np.ascontiguousarray(x[:n // 2])), m(np.ascontiguousarray(x[n // 2:]))])
            bar = mo.DRIFT(n) + (1.5 * mo.RETURN_BAR(n // 2, mo.increment(float(rate))) if algo in ("addfast", "unroll") else 0)
            assert _maxerr(y, g[f"{algo}_r{ri}_y"]) <= bar, (algo, rate)
            m.close()
