"""CPU tests (-m "not gpu") for SURVEY.md §8 row f-4, the PFDSP mixers:
  * the numpy restatement (oracle/pfdsp_oracle.py) is pinned against the REAL reference compiled from
    src/pf_mixer.cpp (oracle/_ref/libpfdsp_ref.so) — outputs, returned phases and state structs;
  * the float64 closed form `exact()` describes every reference algorithm to within its own float drift
    (this is the function the HIP kernels evaluate; the bars used in tests/test_pfdsp.py come from here);
  * libpfdsp_hip.so loads without a GPU, exports every name include/pfdsp_hip.h declares = every name the
    reference exports, and its host-only entries (the *_init / *_update_rate state builders) produce
    bit-identical structs to the reference's.  No mixer kernel is called here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import pfdsp_oracle as mo
from oracle import pfdsp_ref
from pffft_amd import pfdsp


@pytest.fixture(scope="module")
def R():
    if not pfdsp_ref.available():
        from oracle import ref as oref
        oref.build()
    if not pfdsp_ref.available():
        __import__("conftest").missing_checker("oracle/_ref/libpfdsp_ref.so")
    return pfdsp_ref.get()


@pytest.fixture(scope="module")
def H():
    from pffft_amd import build
    build.build()
    return pfdsp.lib()


def _x(n, seed=7):
    rng = np.random.default_rng(seed)
    return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)


def _maxerr(a, b):
    return float(np.abs(np.asarray(a, np.complex128) - np.asarray(b, np.complex128)).max())


RATE, PH0, N = 0.0137, 0.4, 1024


# ------------------------------------------------------------------ restatement pinned to the reference
def test_restated_math_cc(R):
    x = _x(N)
    m = pfdsp.Mixer("math", RATE, PH0, abi=R)
    y = m(x)
    yo, ph = mo.shift_math_cc(x, RATE, PH0)
    assert _maxerr(y, yo) <= 2e-7 and abs(m.phase - float(ph)) <= 1e-6


def test_restated_addfast_unroll(R):
    x = _x(N)
    for algo, fn in (("addfast", mo.shift_addfast_cc), ("unroll", mo.shift_unroll_cc)):
        m = pfdsp.Mixer(algo, RATE, PH0, abi=R)
        y = m(x)
        yo, ph = fn(x, RATE, PH0)
        assert _maxerr(y, yo) <= 4e-7, algo
        assert abs(m.phase - float(ph)) <= 1e-6, algo
        m.close()


def test_restated_limited_unroll(R):
    x = _x(N + 64)                       # one partial block at the end
    m = pfdsp.Mixer("limited_unroll", RATE, abi=R)
    y = m(x)
    yo, st = mo.shift_limited_unroll_cc(x, RATE)
    assert _maxerr(y, yo) <= 4e-7
    assert abs(m.data.complex_phase.i - st[0]) <= 2e-7 and abs(m.data.complex_phase.q - st[1]) <= 2e-7


@pytest.mark.parametrize("variant", ["A", "B", "C"])
def test_restated_limited_unroll_sse(R, variant):
    x = _x(N + 64)
    m = pfdsp.Mixer(f"limited_unroll_{variant}_sse", RATE, PH0, abi=R)
    y = m(x)
    yo, (sc, ss) = mo.shift_limited_unroll_sse(x, RATE, PH0)
    assert _maxerr(y, yo) <= 4e-7
    assert np.abs(np.array(m.data.phase_state_i[:]) - sc).max() <= 3e-7
    assert np.abs(np.array(m.data.phase_state_q[:]) - ss).max() <= 3e-7


@pytest.mark.parametrize("algo,lanes", [("recursive_osc", 8), ("recursive_osc_sse", 4)])
def test_restated_recursive_osc(R, algo, lanes):
    x = _x(N)
    m = pfdsp.Mixer(algo, RATE, PH0, abi=R)
    conf, st = mo.recursive_osc_init(RATE, PH0, lanes)
    assert abs(m.conf.k1 - conf[0]) <= 1e-7 * abs(conf[0]) + 1e-9 and abs(m.conf.k2 - conf[1]) <= 1e-7 * abs(conf[1]) + 1e-9
    assert np.abs(np.array(m.data.u_cos[:lanes]) - st[0]).max() <= 3e-7
    y = m(x)
    yo, st2 = mo.recursive_osc_run(x, (np.float32(m.conf.k1), np.float32(m.conf.k2)),
                                   (np.array(st[0]), np.array(st[1])), lanes)
    assert _maxerr(y, yo) <= 2e-6
    assert np.abs(np.array(m.data.u_cos[:lanes]) - st2[0]).max() <= 2e-6
    if algo == "recursive_osc":
        g = np.empty(64, np.complex64)
        m.generate(g)
        go, _ = mo.recursive_osc_run(np.ones(64, np.complex64), (np.float32(m.conf.k1), np.float32(m.conf.k2)), st2, lanes, gen=True)
        assert _maxerr(g, go) <= 2e-6


# ------------------------------------------------------------------ the closed form describes every algorithm
@pytest.mark.parametrize("algo", pfdsp.ALGOS)
def test_closed_form_describes_reference(R, algo):
    """|reference - exact| stays within DRIFT(n) = 1e-6 + 2e-7 n: the reference accumulates its phase (or its phase
    tables) in float, so its error grows linearly with the stream position — measured 1e-5..3e-5 at n = 256,
    1e-4..5e-4 at 4096, 3e-3..8e-3 at 65536 for the table/accumulator algorithms (A, D-H), ~1e-5 for C, I, J.
    tests/test_pfdsp.py uses the same bound when it compares the HIP output with the reference."""
    if algo == "table":
        pytest.skip("reference shift_table_cc has quadrant resolution only (src/pf_mixer.cpp:202); see test below")
    n = 4096
    x = _x(n, 11)
    m = pfdsp.Mixer(algo, RATE, PH0, abi=R)
    y = m(x)
    rec = algo.startswith("recursive")
    if rec:
        lanes = 8 if algo == "recursive_osc" else 4
        th = mo.osc_step_angle(m.conf.k1, m.conf.k2)
        # lane phasors as initialised, block angle from the float constants
        conf, st = mo.recursive_osc_init(RATE, PH0, lanes)
        i = np.arange(n)
        want = x.astype(np.complex128) * (st[0].astype(np.float64) + 1j * st[1])[i % lanes] * np.exp(1j * th * (i // lanes))
    else:
        inc = mo.increment(RATE)
        want = mo.exact(x, inc, PH0, first=1 if algo == "addfast" else 0)
    assert _maxerr(y, want) <= mo.DRIFT(n), algo
    m.close()


def test_reference_table_quirk(R):
    """shift_table_cc: (int)(vphase/(PI/2)) * table_size is 0 for every phase, so sin/cos come from table[0] /
    table[size-1] only — a 4-level oscillator.  Documented in include/pfdsp_hip.h; not reproduced."""
    x = np.ones(256, np.complex64)
    m = pfdsp.Mixer("table", RATE, 0.0, abi=R, table_size=1024)
    y = m(x)
    assert len(np.unique(np.round(y, 3))) <= 5
    m.close()


# ------------------------------------------------------------------ the product's ABI and host-only entries
def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pfdsp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:shift_|gen_recursive|have_sse|pfdsp_hip_)\w+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(H, R):
    names = declared_symbols()
    assert len(names) == 32, names
    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    mine = exported(pfdsp.lib_path())
    assert not [n for n in names if n not in mine]
    # name for name what the reference's object exports (C linkage), plus the three additive entries
    ref_syms = {s for s in exported(R.path) if not s.startswith("_")}
    assert ref_syms <= mine, sorted(ref_syms - mine)
    assert mine - ref_syms == {"pfdsp_hip_shift_device", "pfdsp_hip_last_error", "pfdsp_hip_error_count"}
    assert set(pfdsp.REFERENCE_ENTRIES) == ref_syms


def _bytes(s):
    return bytes(C.string_at(C.addressof(s), C.sizeof(s)))


def test_state_builders_match_reference_bit_for_bit(H, R):
    assert H.have_sse_shift_mixer_impl() == 1 == R.have_sse_shift_mixer_impl()
    for rate in (0.0137, -0.21, 0.499):
        assert _bytes(H.shift_addfast_init(rate)) == _bytes(R.shift_addfast_init(rate))
        a, b = H.shift_limited_unroll_init(rate), R.shift_limited_unroll_init(rate)
        n = 2 * 128 * 4 + 8 + 4           # the tail of the by-value struct is padding
        assert _bytes(a)[:n] == _bytes(b)[:n]
        for v in "ABC":
            f = f"shift_limited_unroll_{v}_sse_init"
            assert _bytes(getattr(H, f)(rate, 0.3)) == _bytes(getattr(R, f)(rate, 0.3)), f
        for lanes, init, CT, ST in ((8, "shift_recursive_osc_init", pfdsp.shift_recursive_osc_conf_t, pfdsp.shift_recursive_osc_t),
                                    (4, "shift_recursive_osc_sse_init", pfdsp.shift_recursive_osc_sse_conf_t, pfdsp.shift_recursive_osc_sse_t)):
            for ph in (0.0, 1.1):
                ch, sh, cr, sr = CT(), ST(), CT(), ST()
                getattr(H, init)(rate, ph, C.byref(ch), C.byref(sh))
                getattr(R, init)(rate, ph, C.byref(cr), C.byref(sr))
                assert _bytes(ch) == _bytes(cr) and _bytes(sh) == _bytes(sr), (init, rate, ph)
        uh, ur = H.shift_unroll_init(rate, 300), R.shift_unroll_init(rate, 300)
        assert uh.size == ur.size == 300 and uh.phase_increment == ur.phase_increment
        assert np.array_equal(np.ctypeslib.as_array(uh.dsin, (300,)), np.ctypeslib.as_array(ur.dsin, (300,)))
        assert np.array_equal(np.ctypeslib.as_array(uh.dcos, (300,)), np.ctypeslib.as_array(ur.dcos, (300,)))
        H.shift_unroll_deinit(C.byref(uh)); R.shift_unroll_deinit(C.byref(ur))
        assert not uh.dsin and not uh.dcos
    th, tr = H.shift_table_init(512), R.shift_table_init(512)
    assert np.array_equal(np.ctypeslib.as_array(th.table, (512,)), np.ctypeslib.as_array(tr.table, (512,)))
    H.shift_table_deinit(th); R.shift_table_deinit(tr)


# ------------------------------------------------------------------ committed fixtures (work without oracle/_ref)
@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "pfdsp_golden.npz")))


def test_restatement_and_closed_form_against_golden(gold):
    """tests/golden/pfdsp_golden.npz (made by tests/golden/make_pfdsp_golden.py from the reference's object code):
    the numpy restatement reproduces the stored reference outputs of two chained 256-sample calls; the float64 closed
    form stays within DRIFT(512) of them."""
    x, ph0 = gold["x"], float(gold["ph0"][0])
    n = x.size
    for ri, rate in enumerate(gold["rates"]):
        rate = float(rate)
        for algo, fn in (("math", mo.shift_math_cc), ("addfast", mo.shift_addfast_cc), ("unroll", mo.shift_unroll_cc)):
            y0, p = fn(x[:n // 2], rate, ph0)
            y1, p2 = fn(x[n // 2:], rate, p)
            assert _maxerr(np.concatenate([y0, y1]), gold[f"{algo}_r{ri}_y"]) <= 6e-7, (algo, rate)
            assert abs(float(p) - gold[f"{algo}_r{ri}_state_mid"][0]) <= 2e-6
        y0, st = mo.shift_limited_unroll_cc(x[:n // 2], rate, (np.cos(np.float32(ph0)), np.sin(np.float32(ph0))))
        y1, st = mo.shift_limited_unroll_cc(x[n // 2:], rate, st)
        assert _maxerr(np.concatenate([y0, y1]), gold[f"limited_unroll_r{ri}_y"]) <= 6e-7
        for lanes, algo in ((8, "recursive_osc"), (4, "recursive_osc_sse")):
            conf, st = mo.recursive_osc_init(rate, ph0, lanes)
            y0, st = mo.recursive_osc_run(x[:n // 2], conf, st, lanes)
            y1, st = mo.recursive_osc_run(x[n // 2:], conf, st, lanes)
            assert _maxerr(np.concatenate([y0, y1]), gold[f"{algo}_r{ri}_y"]) <= 3e-6, (algo, rate)
        inc = mo.increment(rate)
        chained = 1.5 * mo.RETURN_BAR(n // 2, inc)     # C, D: the second call starts from the phase the first one returned
        for algo in ("math", "unroll", "limited_unroll", "limited_unroll_A_sse", "limited_unroll_B_sse", "limited_unroll_C_sse"):
            bar = mo.DRIFT(n) + (chained if algo == "unroll" else 0)
            assert _maxerr(gold[f"{algo}_r{ri}_y"], mo.exact(x, inc, ph0)) <= bar, (algo, rate)
        assert _maxerr(gold[f"addfast_r{ri}_y"], mo.exact(x, inc, ph0, first=1)) <= mo.DRIFT(n) + chained
