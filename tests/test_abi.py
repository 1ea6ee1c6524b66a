"""CPU tests (-m "not gpu") of the drop-in boundary: the C-ABI library loads without a GPU, exports
every symbol include/pffft_hip.h declares, and its host-only entries (size helpers, setup
validation, aligned allocator) behave like the reference's (tests/test_fft_factors.c:36-61,
tests/test_pffft.c:280-330, src/pffft_common.c:12-43).  No compute entry is called here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import pffft_amd as pa
from conftest import ROOT
from oracle import pffft_oracle as po


@pytest.fixture(scope="module")
def L():
    from pffft_amd import build
    build.build()
    return pa.lib()


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pffft_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b((?:pffftd?_|pffastconv_|validate_pffftd?_)\w+)\s*\(", txt)
    return sorted(set(names))


def test_every_declared_symbol_is_exported(L):
    names = declared_symbols()
    assert len(names) >= 53, names
    out = subprocess.run(["nm", "-D", "--defined-only", pa.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    missing = [n for n in names if n not in exported]
    assert not missing, missing
    # the reference's exported set must be covered name for name
    ref_syms = []
    for pfx in ("pffft", "pffftd"):
        ref_syms += [f"{pfx}_{s}" for s in (
            "new_setup destroy_setup transform transform_ordered zreorder zconvolve_accumulate zconvolve_no_accu "
            "simd_size simd_arch min_fft_size is_valid_size nearest_transform_size next_power_of_two "
            "is_power_of_two aligned_malloc aligned_free").split()]
        ref_syms += [f"validate_{pfx}_simd", f"validate_{pfx}_simd_ex"]
    ref_syms += ["pffastconv_new_setup", "pffastconv_apply", "pffastconv_destroy_setup", "pffastconv_malloc",
                 "pffastconv_free", "pffastconv_simd_size"]
    assert not [s for s in ref_syms if s not in exported]
    # internal helpers of the reference that are hidden there must not leak here either
    assert "pffft_cplx_finalize" not in exported and "pffft_transform_internal" not in exported


def test_introspection(L):
    assert pa.simd_size() == 4 and pa.simd_size(np.float64) == 4  # selects the 4-lane internal layout
    assert pa.simd_arch() == "HIP-gfx950"
    assert L.pffastconv_simd_size() == 4
    assert pa.min_fft_size(pa.REAL) == 32 and pa.min_fft_size(pa.COMPLEX) == 16
    assert L.validate_pffft_simd() == 0 and L.validate_pffftd_simd() == 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_accepted_sizes_match_reference_rules(L, dtype):
    # tests/test_fft_factors.c:36-61: is_valid_size(N) <=> local 2/3/5 factorisation <=> new_setup(N) != NULL
    for tr in (pa.REAL, pa.COMPLEX):
        nmin = pa.min_fft_size(tr, dtype)
        for N in range(nmin // 2, 12 * nmin + 1, nmin // 2):
            assert pa.is_valid_size(N, tr, dtype) == po.is_valid_size(N, tr), (N, tr)
            try:
                s = pa.Setup(N, tr, dtype)
                ok = True
                s.close()
            except ValueError:
                ok = False
            assert ok == po.new_setup_ok(N, tr), (N, tr)
        for N in (1, 17, 100, 1000, 5000, 100000):
            for higher in (False, True):
                assert pa.nearest_transform_size(N, tr, higher, dtype) == po.nearest_transform_size(N, tr, higher)
    for bad in (-1, 0, (1 << 26) + 32, 100, 33):
        with pytest.raises(ValueError):
            pa.Setup(bad, pa.COMPLEX, dtype)


def test_power_of_two_helpers(L):
    for pfx in ("pffft", "pffftd"):
        npo2, ipo2 = getattr(L, pfx + "_next_power_of_two"), getattr(L, pfx + "_is_power_of_two")
        for N in list(range(0, 70)) + [255, 256, 257, 1 << 20, (1 << 20) + 1]:
            assert npo2(N) == po.next_power_of_two(N), N
            assert bool(ipo2(N)) == po.is_power_of_two(N), N


def test_aligned_allocator(L):
    for pfx in ("pffft", "pffftd"):
        m, f = getattr(L, pfx + "_aligned_malloc"), getattr(L, pfx + "_aligned_free")
        ps = [m(n) for n in (1, 17, 4096, 1 << 20)]
        assert all(p and p % 64 == 0 for p in ps)  # src/pffft_common.c:8: 64-byte alignment
        for p in ps:
            C.memset(p, 0xAB, 1)
            f(p)
        f(None)  # NULL-safe
    L.pffastconv_malloc.restype = C.c_void_p
    L.pffastconv_malloc.argtypes = [C.c_size_t]
    L.pffastconv_free.argtypes = [C.c_void_p]
    p = L.pffastconv_malloc(100)
    assert p % 64 == 0
    L.pffastconv_free(p)


def test_fastconv_setup_rules(L):
    # src/pffastconv.c:62-80: Nfft = max(2*next_pow2(filterLen-1), 2*simd^2, next_pow2(blockLen)); CPLX_FILTER -> NULL
    h = np.ones(4096, np.float32)
    fc = pa.FastConv(h, 0, 0)
    assert fc.block_len == 8192
    fc.close()
    fc = pa.FastConv(np.ones(5, np.float32), 0, 0)
    assert fc.block_len == 32
    fc.close()
    fc = pa.FastConv(np.ones(100, np.float32), 1000, 0)
    assert fc.block_len == 1024
    fc.close()
    with pytest.raises(ValueError):
        pa.FastConv(h, 0, 2)
    for taps, blk, flags in ((129, 0, 0), (64, 512, 0), (33, 0, 17)):
        s = po.fastconv_setup(np.ones(taps, np.float32), blk, flags)
        fc = pa.FastConv(np.ones(taps, np.float32), blk, flags)
        assert fc.block_len == s["blockLen"]
        fc.close()


def test_no_oracle_in_product():
    """The shipped package must not import or link the oracle."""
    for root, _, files in os.walk(os.path.join(ROOT, "pffft_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r"(from|import)\s+oracle|#include\s+\"[^\"]*oracle|libpffft_ref|dlopen", txt), f
    out = subprocess.run(["ldd", pa.lib_path()], capture_output=True, text=True).stdout
    assert "pffft_ref" not in out and "fftpack" not in out


_FAILSOFT = r"""
import numpy as np, pffft_amd as pa
s = pa.Setup(64, pa.REAL)
x = np.random.default_rng(0).uniform(-1, 1, 64).astype(np.float32)
assert pa.error_count() == 0
y = s.transform(x)                       # legacy void entry, host pointers, no device
assert np.isnan(y).all() and pa.error_count() == 1 and "no ROCm-capable device" in pa.last_error()
y = s.transform_ordered(x)
assert np.isnan(y).all() and pa.error_count() == 2
fc = pa.FastConv(np.ones(8, np.float32), 0, 0)
yy, n = fc.apply(np.ones(100, np.float32))
assert n == -1 and pa.error_count() == 3      # -1, never 0: a streaming caller must be able to tell failure from "no output yet"
# a C caller that follows the reference's contract allocates inputLen - filterLen + 1 outputs (include/pffft/pffastconv.h:159):
# the fail-soft fill must stay inside them (ADVICE r02: it used to fill inputLen floats)
import ctypes as C
L = pa.lib()
xin = np.ones(100, np.float32); out = np.zeros(100, np.float32)
n = L.pffastconv_apply(fc.handle, xin.ctypes.data, 100, out.ctypes.data, 1)
assert n == -1 and np.isnan(out[:93]).all() and (out[93:] == 0).all(), out
# an invalid handle writes nothing (the vector length would have to be read from the object that failed validation)
z = np.zeros(64, np.float32)
L.pffft_transform(None, x.ctypes.data, z.ctypes.data, None, 0)
assert (z == 0).all() and pa.error_count() == 5
print("FAILSOFT-OK")
"""


def test_legacy_entries_fail_soft_without_a_device():
    """The legacy entries are `void` (include/pffft/pffft.h:159): where the reference could not fail, the drop-in must not
    kill its caller.  Without a HIP device: stderr line + NaN-filled output + pffft_hip_error_count(); abort() only under
    PFFFT_HIP_ABORT=1.  (Runs where no GPU is visible — i.e. in the CPU suite.)"""
    import sys
    if pa.device_count() > 0:
        pytest.skip("a HIP device is present: the failure path cannot be provoked this way")
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("PFFFT_HIP_ABORT", None)
    p = subprocess.run([sys.executable, "-c", _FAILSOFT], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and "FAILSOFT-OK" in p.stdout, p.stdout + p.stderr
    assert "output filled with NaN" in p.stderr
    env["PFFFT_HIP_ABORT"] = "1"
    p = subprocess.run([sys.executable, "-c", _FAILSOFT], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode < 0 and "FAILSOFT-OK" not in p.stdout     # killed by SIGABRT: fail-fast on request only


def test_transform_batch_multi_validates_its_arguments(L):
    """pffft[d]_hip_transform_batch_multi (include/pffft_hip.h; round 5): no parts is a no-op, missing arrays are an error code with a
    message - before any device is touched, so this runs without a GPU.  (The transforms themselves: tests/test_gpu_round5.py.)"""
    for name in ("pffft_hip_transform_batch_multi", "pffftd_hip_transform_batch_multi"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        assert f(0, None, None, None, None, None, 0, 0, None) == 0
        assert f(1, None, None, None, None, None, 0, 0, None) != 0
        assert b"transform_batch_multi" in L.pffft_hip_last_error()
        assert f(-1, None, None, None, None, None, 0, 0, None) != 0


def test_legal_size_enumeration_matches_the_library():
    """tests/test_gpu_round3.py walks conftest.legal_sizes(): exactly what the drop-in's is_valid_size accepts, which is
    the reference's set (tests/test_fft_factors.c:36-61; the reference program itself runs in test_reference_programs)."""
    from conftest import legal_sizes
    L = pa.lib()
    for tr in (pa.REAL, pa.COMPLEX):
        nmin = 32 if tr == pa.REAL else 16
        mine = set(legal_sizes(tr, 0, 1 << 16))
        assert mine == {N for N in range(nmin, (1 << 16) + 1, nmin) if L.pffft_is_valid_size(N, tr)}
        for N in sorted(mine)[::9]:
            s = L.pffft_new_setup(N, tr)
            assert s, N
            L.pffft_destroy_setup(s)


def test_tile_planner_over_every_legal_size():
    """The planner of the tile passes beyond LDS (tile_tu.hip, reached through pffft_hip_tile_plan: host arithmetic, no GPU): for
    EVERY legal core size n = 16 2^a 3^b 5^c up to 2^24, both precisions, with and without `deep`, a plan is 2 or 3 tile lengths
    whose product is n, each one an instantiated length: power of two 64 .. 1024, R0 2^b with R0 in {3, 5, 9, 15, 25, 27, 45}, 48 .. 768
    (register-tiled, fft_tile.h), or - round 4 - any 2^a 3^b 5^c from 32 to 864 on the run-time plans of fft_tileg.h (float: even, a 16-byte
    unit is two sequences of the other pass; double next to a register-tiled pass: a multiple of 8, those kernels are built without the
    ragged last tile).  A plan that names a length without a kernel would only fail at launch time, beyond the sizes the GPU walk of
    tests/test_gpu_round3.py covers.  With the run-time lengths EVERY legal size beyond LDS up to 2^21 has a plan when the streaming route
    would take five sweeps (deep)."""
    import pffft_amd as pa
    from conftest import legal_sizes

    def register_tiled(L):
        r0 = L
        b = 0
        while r0 % 2 == 0:
            r0 //= 2; b += 1
        if r0 == 1:
            return 6 <= b <= 10
        if r0 not in (3, 5, 9, 15, 25, 27, 45) or not 48 <= L <= 768:
            return False
        return b >= (3 if r0 >= 9 else 4)

    def run_time(L, is_double):
        m = L
        for q in (2, 3, 5):
            while m % q == 0:
                m //= q
        return m == 1 and 32 <= L <= 864 and (is_double or L % 2 == 0)

    covered = {False: 0, True: 0}
    sizes = [n for n in legal_sizes(pa.COMPLEX, 16, 1 << 24)]
    assert len(sizes) > 500
    for is_double in (False, True):
        for n in sizes:
            for deep in (False, True):
                plan = pa.tile_plan(n, is_double, deep)
                if not plan:
                    continue
                assert len(plan) in (2, 3), (n, plan)
                prod = 1
                for L in plan:
                    prod *= L
                    assert register_tiled(L) or run_time(L, is_double), (n, is_double, deep, plan)
                assert prod == n, (n, plan)
                if is_double and len(plan) == 2:       # a register-tiled double pass sees whole tiles of 8 sequences
                    for L, other in ((plan[0], plan[1]), (plan[1], plan[0])):
                        # (a length that has both kernels - 720 next to 810 - runs on its run-time plan there: pffft_hip_tile_plan reports
                        #  lengths only, the planner's own legality rule tile_pair_legal() holds the kernel choice to this)
                        assert not register_tiled(L) or other % 8 == 0 or run_time(L, is_double), (n, plan)
                if n & (n - 1):
                    assert all(L <= 864 for L in plan) and (len(plan) == 2 or deep), (n, deep, plan)
                if deep and n & (n - 1):
                    covered[is_double] += 1
            # without `deep` a plan is never costlier than with it: whatever is planned shallow is planned deep as well
            if pa.tile_plan(n, is_double, False):
                assert pa.tile_plan(n, is_double, True), n
            if n <= (1 << 21) and n * (16 if is_double else 8) > 80000 and n & (n - 1):
                assert pa.tile_plan(n, is_double, True), (n, is_double)
    # the sizes DESIGN.md §3.5 names.  Round 5: where tools/tune_tile_plans.py MEASURED a pair faster than the cost model's choice, the table
    # tile_plan_gen.h decides (61440: 256 x 240 -> 480 x 128 +3 %, 9216 double: 64 x 144 -> 128 x 72 +11 %, 12000: 100 x 120 -> 200 x 60 +21 %)
    assert pa.tile_plan(61440) == [480, 128] and pa.tile_plan(115200) == [480, 240] and pa.tile_plan(9216, True) == [128, 72]
    assert pa.tile_plan(1024000) == [] and len(pa.tile_plan(1024000, False, True)) == 3
    # a complex-transform plan from the table does not plan the core of a REAL transform where the model plans none (mode 2)
    assert pa.tile_plan(12000) == [200, 60] and pa.tile_plan(12000, False, 2) == [] and pa.tile_plan(12000, True) == []
    assert pa.tile_plan(288000) == [480, 600]                    # (round 3: [] / [400, 720] - 600 = 75 x 8 is a run-time length)
    assert pa.tile_plan(518400, False, 2) == [] and pa.tile_plan(518400, False, True) == [600, 864]      # two costly passes beat five sweeps
    # float complex, three streaming sweeps: run-time lengths that carry the internal layout are taken up to a wider cost bar (mode 0), not for
    # the core of a real transform (mode 2); in double the model keeps the streaming route, the measurement prefers two run-time passes
    # (0.242 / 0.253 / 0.245 / 0.186 -> 0.220 / 0.236 / 0.243 / 0.242: the backward unordered transform no longer pays a layout sweep)
    assert pa.tile_plan(10800) == [180, 60] and pa.tile_plan(10800, False, 2) == [] and pa.tile_plan(10800, True) == [180, 60]
    assert pa.tile_plan(600000) == [] and pa.tile_plan(600000, False, True) == [750, 800] and pa.tile_plan(314928, True, True) == [486, 648]
    assert pa.tile_plan(1 << 16) == [512, 128] and pa.tile_plan(1 << 18) == [512, 512] and pa.tile_plan(1 << 22) == [128, 128, 256] and pa.tile_plan(2048) == []
    # round 5: column tiles of L >= 576 spill (168 VGPRs for nine and more wavefronts): priced, N = 82944 no longer 576 x 144
    assert pa.tile_plan(82944, True)[0] < 576 and pa.tile_plan(294912, True) == [512, 576]
    assert covered[False] > 100 and covered[True] > 150, covered



def test_describe_matches_the_routing_restated_here():
    """pffft_hip_describe() prints the Route records the planner stored in the setup at pffft_new_setup (round 5: the decisions used to be
    re-derived on every call).  Walk over EVERY legal size up to 2^18 and every 7th up to 2^21, both precisions, real and complex: the
    family line equals the restatement of tests/test_gpu_round3.py, and per (direction, layout): which power-of-two
    sizes run their Stockham plan instead of the register-tiled kernel (DESIGN.md §3.4 routing, restated), launch rules, and beyond LDS
    the core - tile lengths equal to pffft_hip_tile_plan's, the two-sweep real route for double N = 2^18 / 2^19 forward only, the
    internal layout fused into the first / last pass exactly where the plan's lengths are multiples of 4.  No device needed."""
    from conftest import legal_sizes
    from test_gpu_round3 import expected_family

    def prefers_stock(n, cplx, fwd, dbl):
        if not dbl:
            return (n <= 64 or n in (128, 8192)) if cplx else (n <= 64 or (n == 8192 and not fwd))
        return (n <= 256 or n >= 2048) if cplx else (n <= 64 or (n == 128 and not fwd) or n == 256 or n >= 2048)

    seen = {}
    for dt in ("f32", "f64"):
        dbl = dt == "f64"
        dtype = np.float64 if dbl else np.float32
        for tr in (pa.COMPLEX, pa.REAL):
            sizes = legal_sizes(tr, 0, 1 << 18) + legal_sizes(tr, (1 << 18) + 1, 1 << 21)[::7]
            for N in sizes:
                s = pa.Setup(N, tr, dtype)
                text = pa.describe(s)
                lines = text.strip().split("\n")
                assert len(lines) == 5, text
                fam = expected_family(N, tr, dt)
                assert lines[0].endswith("family " + fam) and pa.kernel_name(s) == fam, (dt, tr, N, lines[0])
                n = N if tr == pa.COMPLEX else N // 2
                for ln in lines[1:]:
                    head, body = ln.strip().split(": ", 1)
                    fwd, ordered = head.startswith("forward"), "unordered" not in head
                    kind = body.split(":")[0]
                    seen[kind] = seen.get(kind, 0) + 1
                    if fam in ("tiny", "c1024_f32", "stockham"):
                        assert kind == fam, (dt, tr, N, ln)
                        if fam == "c1024_f32":
                            assert "<= 4 resident sets" in body
                    elif fam == "tiled":
                        # (a Stockham plan exists where two exchange images fit LDS: n * sizeof(complex) <= 80 000 B)
                        has_plan = n * (16 if dbl else 8) <= 80000
                        want = "stockham" if has_plan and prefers_stock(n, tr == pa.COMPLEX, fwd, dbl) else "tiled"
                        assert kind == want, (dt, tr, N, ln)
                        if kind == "tiled":
                            assert "in-order oneshot<=" + ("16" if (not dbl and n == 4096 and tr == pa.COMPLEX) else "4") in body, ln
                    elif n * (16 if dbl else 8) <= 147456:
                        # round 6: the vectors that fill LDS once but not twice (80 000 B < vector <= 144 KiB) take ONE pass on the single-image
                        # kernel (fft_one.h) in all four combinations; the size's family stays "fourstep" (its helpers run those passes)
                        assert kind == "oneimage" and "1 sweep" in body and "in-order" in body, (dt, tr, N, ln)
                        rad = [int(v) for v in body.split("stages ")[1].split(" in place")[0].split(" x ")]
                        prod = 1
                        for v in rad:
                            prod *= v
                        assert prod == n and 2 <= len(rad) <= 4 and rad[0] >= 8 and rad[-1] >= 8, (dt, tr, N, ln)
                    else:
                        assert kind == "fourstep", (dt, tr, N, ln)
                        if "real-rows" in body:       # round 6: the pair pass inside the last tile pass - real forward ordered on two-pass plans
                            assert tr == pa.REAL and fwd and ordered and "tiles " in body and " 2 sweeps" in body, ln
                            assert "post -1" in body and "pair_after 0" in body, ln
                        two_sweep = tr == pa.REAL and dbl and N in (1 << 18, 1 << 19) and fwd
                        assert ("real two-sweep" in body) == two_sweep, (dt, tr, N, ln)
                        if "tiles " in body and not two_sweep:
                            lens = [int(v) for v in body.split("tiles ")[1].split(" (mode")[0].split(" x ")]
                            mode = int(body.split("(mode ")[1][0])
                            assert lens == pa.tile_plan(n, dbl, mode), (dt, tr, N, ln)
                            prod = 1
                            for v in lens:
                                prod *= v
                            assert prod == n
                            # the layout is blocks of four adjacent bins of the four spectrum quarters: the pass that reads / stores it needs
                            # its tile length (the quarters) and its sequence count (the product of the other lengths) in multiples of 4
                            out_ok = all(v % 4 == 0 for v in lens)
                            in_ok = lens[0] % 4 == 0 and (n // lens[0]) % 4 == 0 and (len(lens) == 3 or lens[1] % 4 == 0)
                            if tr == pa.COMPLEX and not ordered:
                                assert ("fuse_out 1" in body) == (fwd and out_ok), ln
                                assert ("fuse_in 1" in body) == ((not fwd) and in_ok), ln
                        if tr == pa.COMPLEX and ordered:
                            assert "pre -1" in body and "post -1" in body and any(f" {k} sweeps" in body for k in (2, 3, 5)), ln
                s.close()
    assert seen.get("tiled", 0) > 60 and seen.get("stockham", 0) > 800 and seen.get("fourstep", 0) > 1900 and seen.get("tiny", 0) >= 8 and seen.get("oneimage", 0) == 4 * 62, seen
    # an invalid handle
    import ctypes as C
    buf = C.create_string_buffer(64)
    assert pa.lib().pffft_hip_describe(None, buf, 64) == -1 and buf.value == b""
    # truncation: snprintf convention
    s = pa.Setup(1024, pa.COMPLEX)
    need = pa.lib().pffft_hip_describe(s.handle, buf, 64)
    assert need > 64 and len(buf.value) == 63 and pa.describe(s).startswith(buf.value.decode())
    s.close()


def test_setup_devices_without_a_device(L):
    """pffft_hip_setup_devices (round 6: one setup, any device - the device state of a setup is kept per device and listed here): no state
    before the first transform, an invalid handle lists nothing, and a legacy call that fails soft for want of a GPU leaves the list empty
    (nothing was built, nothing to release) - destroy_setup afterwards is clean."""
    buf = (C.c_int * 4)(9, 9, 9, 9)
    assert L.pffft_hip_setup_devices(None, buf, 4) == 0 and list(buf) == [9, 9, 9, 9]
    for dtype in (np.float32, np.float64):
        s = pa.Setup(1024, pa.COMPLEX, dtype)
        assert pa.setup_devices(s) == []
        assert L.pffft_hip_setup_devices(s.handle, None, 0) == 0
        s.close()
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        e0 = pa.error_count()
        s = pa.Setup(64, pa.REAL)
        y = s.transform(np.ones(64, np.float32), pa.FORWARD)     # fails soft: NaN output + counter (include/pffft_hip.h)
        assert np.isnan(y).all() and pa.error_count() == e0 + 1
        assert pa.setup_devices(s) == []
        s.close()
