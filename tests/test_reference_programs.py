"""The reference's OWN test / validation / example programs (tests/test_pffft.c, test_pffft.cpp,
test_fft_factors.c, test_pffastconv.c, benchmarks/bench_pffft.c --validate, examples/*), compiled from their
sources against libpffft_hip.so by tests/refprogs/Makefile ("the cheapest, strongest
proof of drop-in").  The executables are built in the dev container (where /root/reference exists) into
tests/_refbin/ and travel to the GPU box with the snapshot; the tests skip when they are absent on a CPU-only box and FAIL when they are absent on a GPU box."""
import os
import subprocess

import pytest

from conftest import ROOT

BIN = os.path.join(ROOT, "tests", "_refbin")


def _run(name, *args, timeout=900):
    exe = os.path.join(BIN, name)
    if not os.path.exists(exe):
        from conftest import missing_checker
        missing_checker(exe)
    p = subprocess.run([exe, *args], capture_output=True, text=True, timeout=timeout)
    return p.returncode, p.stdout + p.stderr


def test_fft_factors_program():
    """tests/test_fft_factors.c: accepted-size set == local 2/3/5 factorisation == new_setup != NULL.
    Needs no GPU (setups are host-side plans): runs in the CPU suite."""
    rc, out = _run("test_fft_factors")
    assert rc == 0, out[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("prog,ok", [("test_pffft_float", "all tests succeeded successfully"),
                                     ("test_pffft_double", "all tests succeeded successfully")])
def test_pffft_program(prog, ok):
    """tests/test_pffft.c: N = 32..65536, real+complex, ordered+unordered: single-tone dynamic range >= 140 dB
    (float) / 215 dB (double), phase, magnitude, round trip — through the legacy host-pointer entries."""
    rc, out = _run(prog)
    assert rc == 0 and ok in out, out[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("prog", ["test_pffft_cpp", "test_pffft_cpp11"])
def test_pffft_cpp_program(prog):
    """tests/test_pffft.cpp through the header-only pffft.hpp wrapper (C++98 and C++11 builds)."""
    rc, out = _run(prog)
    assert rc == 0, out[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("prog,sel", [("bench_pffft_float", "--cplx"), ("bench_pffft_float", "--real"),
                                      ("bench_pffft_double", "--cplx"), ("bench_pffft_double", "--real")])
def test_bench_pffft_validate(prog, sel):
    """benchmarks/bench_pffft.c:292-455 --validate: forward vs FFTPACK, in-place == out-of-place bit-exactly,
    zreorder round trip, inverse, zconvolve identity, for the radix-2/3/4/5 size list (:445).
    (Selector first: `--validate` returns immediately, SURVEY.md Appendix C.)"""
    rc, out = _run(prog, sel, "--validate")
    assert rc == 0 and "successful" in out.lower(), out[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("--no-bench", "--quick"), ("--no-bench", "--quick", "--sym")])
def test_pffastconv_program(args):
    """tests/test_pffastconv.c (ctest: test_pfconv_lens_*): output lengths of every block size against the
    naive FIR, real / complex 2xFFT / complex single-FFT."""
    rc, out = _run("test_pffastconv", *args, timeout=1800)
    assert rc == 0, out[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("prog", ["example_c_real_flt_fwd", "example_c_cplx_dbl_fwd", "example_cpp98_real_flt_fwd",
                                  "example_cpp11_cplx_dbl_fwd", "example_cpp98_cplx_flt_fwd", "example_cpp11_real_dbl_fwd"])
def test_examples(prog):
    """All six programs of the reference's examples/ directory (examples/CMakeLists.txt)."""
    rc, out = _run(prog)
    assert rc == 0, out[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("prog,args", [("test_pffastconv", ("--no-len", "--quick", "--sym")),       # ctest bench_pfconv_symetric
                                       ("test_pffastconv", ("--no-len", "--quick")),                # ctest bench_pfconv_non_sym
                                       ("bench_pffft_float", ("--max-len", "128", "--quick")),      # ctest bench_pffft_pow2
                                       ("bench_pffft_float", ("--non-pow2", "--max-len", "192", "--quick"))])   # ctest bench_pffft_non2
def test_reference_bench_mode_ctest_lines(prog, args):
    """The four bench-mode lines of the reference's ctest list (tests/CMakeLists.txt:144-161): the programs time themselves with
    clock() and assert nothing upstream beyond their exit code; here they run the legacy host-pointer entries of the drop-in
    (one launch + one synchronisation per call: the rates they print are the PCIe / launch-inclusive ones of include/pffft_hip.h)."""
    rc, out = _run(prog, *args, timeout=1800)
    assert rc == 0, out[-3000:]


@pytest.mark.gpu
def test_bench_mixers_program():
    """benchmarks/bench_mixers.cpp — the reference's only program over pf_mixer.h — built from its source against
    libpfdsp_hip.so: the nine benches its source enables (shift_math_cc, gen_recursive_osc_c as signal generator, every
    in-place algorithm C..J, state structs by value / by pointer) run 1 MSample in 64 Ki blocks through the legacy
    host-pointer entries.  The program reports rates only (from clock(), i.e. host CPU time: meaningless here); values
    are checked in tests/test_pfdsp.py."""
    rc, out = _run("bench_mixers", "65536", "1", timeout=600)
    assert rc == 0, out[-3000:]
    for name in ("shift_math_cc", "shift_addfast_inp_c", "shift_unroll_inp_c", "shift_limited_unroll_inp_c",
                 "shift_limited_unroll_A_sse_inp_c", "shift_limited_unroll_B_sse_inp_c", "shift_limited_unroll_C_sse_inp_c",
                 "shift_recursive_osc_cc", "shift_recursive_osc_sse_c"):
        assert f"starting bench of {name}" in out, name
    assert out.count("processed 0.983040 Msamples") == 9, out[-3000:]
