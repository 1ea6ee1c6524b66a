"""Measure, for every legal complex size beyond LDS (powers of two up to 2^18 included), EVERY legal pair of tile lengths (pffft_hip_tile_candidates)
and the streaming route, in the four direction x layout combinations, and write the plans that beat the cost model's choice to
pffft_amd/csrc/tile_plan_gen.h (development tool, runs on the GPU):

    python tools/tune_tile_plans.py [HI=600000] [f32|f64|both] [MiB per launch=512]

Per candidate: pffft_hip_tile_override -> new setup -> 6 untimed + 12 timed launches per combination; the model's plan and the two best
challengers are then timed twice more, interleaved, and averaged (a single run scatters by ~3 %).  Score = mean of the four fractions of 8 TB/s;
a challenger replaces the model's plan when its score is >= 3 % better and its minimum is not lower by more than 2 %, or when it lifts a
minimum below 0.20 by >= 10 % without losing more than 2 % of the score.  The values of every candidate are
checked against a float64 DFT of one vector (a plan that computes garbage must never enter the table).  Log: profiles/r05_tile_plan_tuning.txt."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pffft_amd as pa  # noqa: E402
from size_scan import legal_sizes  # noqa: E402

L = pa.lib()
L.pffft_hip_tile_candidates.restype = C.c_int
L.pffft_hip_tile_candidates.argtypes = [C.c_longlong, C.c_int, C.POINTER(C.c_int), C.c_int]
L.pffft_hip_tile_override.restype = C.c_int
L.pffft_hip_tile_override.argtypes = [C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]



def box_stamp():
    """One comment line naming the box a measured table was taken on (device, CU count, ROCm / HIP version, date): the tables are relative A/B
    figures of ONE box - boxes of the pool differ by +-1.5-3 % (Stockham family up to 6 %) - and say which."""
    import datetime
    import torch
    p = torch.cuda.get_device_properties(0)
    return (f"// Taken on: {p.name} ({getattr(p, 'gcnArchName', '?')}, {p.multi_processor_count} CUs), HIP {torch.version.hip}, "
            f"{datetime.datetime.utcnow().strftime('%Y-%m-%d')}; relative A/B figures of one box.\n")


def candidates(n, dbl):
    buf = (C.c_int * (5 * 64))()
    cnt = min(64, L.pffft_hip_tile_candidates(n, int(dbl), buf, 64))
    return [tuple(buf[5 * i: 5 * i + 5]) for i in range(cnt)]


def measure(n, dt, mib):
    """fractions of 8 TB/s [fwd ord, fwd uno, bwd ord, bwd uno] of a fresh setup of n under the current override + value check"""
    tdt = torch.float64 if dt == np.float64 else torch.float32
    isz = np.dtype(dt).itemsize
    s = pa.Setup(n, pa.COMPLEX, dt)
    batch = max(2, (mib << 20) // (s.vec_scalars * isz))
    x = torch.rand(batch, s.vec_scalars, device="cuda", dtype=tdt) * 2 - 1
    y = torch.empty_like(x)
    res = []
    for d in (pa.FORWARD, pa.BACKWARD):
        for o in (True, False):
            for _ in range(6): s.transform_batch(x, y, d, o)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(12): s.transform_batch(x, y, d, o)
            b.record(); torch.cuda.synchronize()
            res.append(2 * x.numel() * isz / (a.elapsed_time(b) / 12 * 1e-3) / 8e12)
    xs = x[:1].contiguous()
    fo = s.transform_batch(xs, None, pa.FORWARD, True)[0].cpu().numpy().astype(np.float64)
    fu = s.transform_batch(xs, None, pa.FORWARD, False)
    xh = xs[0].cpu().numpy().astype(np.float64)
    want = np.fft.fft(xh[0::2] + 1j * xh[1::2])
    err = float(np.abs((fo[0::2] + 1j * fo[1::2]) - want).max() / np.abs(want).max())
    ok = err <= (1e-12 if dt == np.float64 else 1e-5)
    ok &= bool(torch.equal(s.zreorder_batch(fu, None, pa.FORWARD)[0].cpu(), torch.from_numpy(fo.astype(dt))))
    back = s.transform_batch(fu, None, pa.BACKWARD, False)
    ok &= float((back / n - xs).abs().max()) <= (8e-12 if dt == np.float64 else 8e-5)
    desc = pa.describe(s).split("\n")[1].split("fourstep: ")[1].split(";")[0]
    s.close(); del x, y
    return res, ok, desc


def main():
    explicit = None
    if len(sys.argv) > 1 and sys.argv[1] == "sizes":              # tune these sizes only; the other entries of the table are kept
        explicit = [int(v) for v in sys.argv[2].split(",")]
        del sys.argv[1]
        sys.argv[1] = str(max(explicit))
    hi = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
    which = sys.argv[2] if len(sys.argv) > 2 else "both"
    mib = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    table, log = [], []
    kept = []
    if explicit is not None:
        import re
        src = open(os.path.join(ROOT, "pffft_amd", "csrc", "tile_plan_gen.h")).read()
        for m in re.finditer(r"\{(\d+), (\d), (\d+), (\d), (\d+), (\d)\},", src):
            e = tuple(int(v) for v in m.groups())
            if e[0] and e[0] not in explicit:
                kept.append(e)
    for dt in ([np.float32, np.float64] if which == "both" else [np.float64 if which == "f64" else np.float32]):
        dbl = dt == np.float64
        for n in (sorted(explicit) if explicit is not None else legal_sizes(pa.COMPLEX, 2048, hi)):
            if n & (n - 1) == 0 and n > (1 << 18):
                continue                                            # (2^19, 2^20: a 1024-point tile on one side, no candidates; beyond: three passes)
            s = pa.Setup(n, pa.COMPLEX, dt)
            fam = pa.kernel_name(s); s.close()
            if fam != "fourstep":
                continue
            L.pffft_hip_tile_override(n, int(dbl), -1, 0, 0, 0)
            _, _, desc0 = measure(n, dt, mib)                       # the model's plan (this first run also warms the clocks: not scored)
            rows = []
            for (l1, g1, l2, g2, cost) in candidates(n, dbl):
                if L.pffft_hip_tile_override(n, int(dbl), l1, g1, l2, g2) != 0:
                    continue
                r, ok, desc = measure(n, dt, mib)
                rows.append([f"{l1}{'g' if g1 else ''}x{l2}{'g' if g2 else ''} cost {cost}", (l1, g1, l2, g2), r, ok, desc])
            L.pffft_hip_tile_override(n, int(dbl), 0, 0, 0, 0)       # the streaming passes as a candidate (or as the model's own choice)
            r, ok, desc = measure(n, dt, mib)
            if "tiles" not in desc:
                rows.append(["streaming", (0, 0, 0, 0), r, ok, desc])
            base_row = next((q for q in rows if q[4] == desc0), None)
            if base_row is None:
                L.pffft_hip_tile_override(n, int(dbl), -1, 0, 0, 0)
                print(f"?? {np.dtype(dt).name} n={n}: the model's plan [{desc0}] is not among the candidates", flush=True)
                continue
            for q in rows:
                if not q[3]:
                    log.append(f"!! {np.dtype(dt).name} n={n} {q[0]}: WRONG VALUES ({q[4]})")
            # second round: the model's plan and the two best challengers once more, interleaved (a single run scatters by ~3 %)
            chall = sorted((q for q in rows if q is not base_row and q[3]), key=lambda q: -float(np.mean(q[2])))[:2]
            for rep in range(2):
                for q in [base_row] + chall:
                    L.pffft_hip_tile_override(n, int(dbl), *q[1]) if q[1] != (0, 0, 0, 0) or q[0] == "streaming" else None
                    r2, ok2, _ = measure(n, dt, mib)
                    q[2] = [0.5 * (u + v) for u, v in zip(q[2], r2)] if rep == 0 else [(2 * u + v) / 3 for u, v in zip(q[2], r2)]
                    q[3] = q[3] and ok2
            L.pffft_hip_tile_override(n, int(dbl), -1, 0, 0, 0)
            base = base_row[2]
            sc0, mn0 = float(np.mean(base)), min(base)
            best = None
            for name, plan, r, ok, desc in chall:
                if not ok:
                    continue
                sc, mn = float(np.mean(r)), min(r)
                better = (sc >= 1.03 * sc0 and mn >= 0.98 * mn0) or (mn0 < 0.20 and mn >= 1.10 * mn0 and sc >= 0.98 * sc0)
                if better and (best is None or sc > best[0]):
                    best = (sc, mn, name, plan, r, desc)
            line = f"{np.dtype(dt).name} n={n:7d} model [{desc0}] " + " ".join(f"{v:.3f}" for v in base) + f" (mean {sc0:.3f})"
            if best:
                line += f"  ->  {best[2]} [{best[5]}] " + " ".join(f"{v:.3f}" for v in best[4]) + f" (mean {best[0]:.3f})"
                table.append((n, int(dbl)) + tuple(best[3]))
            log.append(line)
            print(line, flush=True)
            for name, plan, r, ok, desc in rows:
                print(f"      {name:28s} " + " ".join(f"{v:.3f}" for v in r) + f"  mean {np.mean(r):.3f}{'' if ok else '  WRONG VALUES'}", flush=True)
            torch.cuda.empty_cache()
    out = os.path.join(ROOT, "gpurun_out", "tile_plan_gen.h")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        f.write("// GENERATED by tools/tune_tile_plans.py - tile plans beyond LDS that were measured to beat the choice of the cost model (tile_tu.hip tile_cost)\n"
                "// on MI355X by >= 3 % of the mean over the four direction x layout combinations (or to lift a minimum below 0.20 by >= 10 %), "
                f"{mib} MiB per launch.\n// {{n, is_double, L1, gen1, L2, gen2}}: columns L1 then rows L2, gen = the run-time kernel of fft_tileg.h; L1 = 0: no tile plan, the\n"
                "// streaming passes of fft_big.h win.  Log of the run: profiles/r05_tile_plan_tuning.txt\n" + box_stamp() + "#pragma once\nnamespace pf {\n"
                "struct TilePlanEnt { long long n; int is_double, l1, g1, l2, g2; };\nstatic const TilePlanEnt kTilePlans[] = {\n")
        for e in sorted(kept + table, key=lambda e: (e[1], e[0])):
            f.write("    {%d, %d, %d, %d, %d, %d},\n" % e)
        f.write("    {0, 0, 0, 0, 0, 0}};\n}  // namespace pf\n")
    with open(os.path.join(ROOT, "gpurun_out", "tile_plan_tuning.txt"), "w") as f:
        f.write("\n".join(log) + "\n")
    print(f"# {len(table)} plans written to {out}")


if __name__ == "__main__":
    main()
