"""Host-side hygiene of the generated sources (no GPU): the compile-time Stockham plans under pffft_amd/csrc/ must be exactly
what tools/gen_stock_plans.hip writes from the committed planner (stock_plan.h), and the measured adoption table of the
direct-first-stage kernels (stock_df_gen.h) must be well-formed - a planner edit without regeneration would silently send
every non-power-of-two size to the slow run-time-plan kernel (the launcher matches plans by memcmp)."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "pffft_amd", "csrc")


def test_stockham_plans_are_in_sync_with_the_planner(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not present")
    exe = tmp_path / "gen_stock_plans"
    subprocess.run([hipcc, "-std=c++17", "-O1", "--offload-arch=gfx950", os.path.join(ROOT, "tools", "gen_stock_plans.hip"),
                    "-o", str(exe)], check=True, capture_output=True, timeout=600)
    out = tmp_path / "gen"
    out.mkdir()
    subprocess.run([str(exe), str(out)], check=True, capture_output=True, timeout=120)
    for tag in (t + "_" + h for t in ("f32c", "f32r", "f64c", "f64r") for h in "ab"):
        name = f"stock_ct_{tag}_gen.hip"
        assert (out / name).read_bytes() == open(os.path.join(CSRC, name), "rb").read(), \
            f"{name} is stale: run tools/tune_stock.sh"


def test_direct_first_stage_table_is_well_formed():
    txt = open(os.path.join(CSRC, "stock_df_gen.h")).read()
    m = re.search(r"kStockDf\[\] = \{([^}]*)\}", txt)
    assert m
    keys = [int(v.strip().rstrip("u")) for v in m.group(1).split(",") if v.strip()]
    assert keys[-1] == 0 and keys[:-1] == sorted(set(keys[:-1])), "sorted, unique, zero-terminated"
    for k in keys[:-1]:
        n, out_int, bwd, real = k & 0xFFFFF, (k >> 20) & 1, (k >> 21) & 1, (k >> 22) & 1
        assert 16 <= n <= 20480 and n % 16 == 0
        assert not (out_int and bwd), "the internal layout is the OUTPUT of forward transforms only"
        assert k >> 24 == 0


def test_every_lds_resident_size_has_a_fast_kernel():
    """Routing (no GPU needed: plans are host data).  Every legal non-power-of-two size whose Stockham images fit LDS must find
    its compile-time plan ("stockham", not the run-time-plan twin "stockham_rt" at 0.3 of the roofline and less), and the
    sizes beyond must take the streaming passes ("fourstep") rather than the in-place radix 2-5 kernel whenever they factor
    into a register-sized radix times a fast row size."""
    import numpy as np
    import pffft_amd as pa

    def smooth5(m):
        for q in (2, 3, 5):
            while m % q == 0:
                m //= q
        return m == 1

    slow = []
    for dt in (np.float32, np.float64):
        for tr, mul in ((pa.COMPLEX, 1), (pa.REAL, 2)):
            for n in range(48, 10241, 16):
                if not smooth5(n) or (n & (n - 1)) == 0:
                    continue
                s = pa.Setup(n * mul, tr, dt)
                name = pa.kernel_name(s)
                s.close()
                if name not in ("stockham", "fourstep", "tiny", "tiled"):
                    slow.append((np.dtype(dt).name, tr, n * mul, name))
    # sizes with too few small factors for four Stockham stages AND no R x N2 factorization (3^5 / 3^6 multiples): known, rare
    known = {3888, 7776, 5832, 11664, 9720, 19440, 15552, 2 * 3888, 2 * 7776}
    assert [x for x in slow if x[2] not in known] == [], slow
    for N, tr, want in ((12000, pa.COMPLEX, "fourstep"), (20480, pa.COMPLEX, "fourstep"), (36864, pa.REAL, "fourstep"),
                        (8000, pa.REAL, "stockham"), (9600, pa.COMPLEX, "stockham")):
        s = pa.Setup(N, tr, np.float32)
        assert pa.kernel_name(s) == want, (N, tr, pa.kernel_name(s))
        s.close()
