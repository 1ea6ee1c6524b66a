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
    for tag in ("f32c", "f32r", "f64c", "f64r"):
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
