"""python tools/pmc_round.py <round>   (right after `bash tools/profile_round.sh <round>` on the GPU box, on the same sources)
Condense gpurun_out/prof_r<round> into profiles/r<round>_*: kernel stats of the bench command, and per
dominant kernel of every BASELINE config: average duration, algorithmic bytes, HBM traffic from PMC
(2 x FETCH_SIZE KiB + WRITE_SIZE KiB, separate passes; MI355X_MICROARCH.md §HBM), LDS conflict ratio, instruction counts."""
import collections, csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "06"
G = os.path.join(ROOT, "gpurun_out", "prof_r" + RND)
P = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(G, "bench", "trace_kernel_stats.csv"), os.path.join(P, f"r{RND}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(G, "cfg", "trace_kernel_stats.csv"), os.path.join(P, f"r{RND}_configs_kernel_stats.csv"))
# durations from the kernel trace of the un-instrumented pass; only the dispatches with the kernel's LARGEST grid count
# (the same kernel also runs once on a single vector when a pffastconv setup transforms its filter)
trace = list(csv.DictReader(open(os.path.join(G, "cfg", "trace_kernel_trace.csv"))))
def big_dispatches(rows, sub, name_key, grid_key="Grid_Size", part=None, wgs=None):
    import re
    m = [r for r in rows if (re.search(sub[3:], r[name_key]) if sub.startswith("re:") else sub in r[name_key])]
    if not m:
        return []
    g = max(int(r[grid_key]) for r in m)
    if wgs is not None:             # a kernel launched with several grids: the dispatches of `wgs` workgroups
        wsz = int(m[0].get("Workgroup_Size", m[0].get("Workgroup_Size_X", 256)))
        g = wgs * wsz
    m = [r for r in m if int(r[grid_key]) == g]
    if part is not None:            # first / second half of the dispatches (two workloads on one kernel, launched one after the other)
        h = len(m) // 2
        m = m[:h] if part == 0 else m[h:]
    return m
PART = {"c4_long": 0, "c4_batch": 1}
# (key, kernel-name substring, algorithmic bytes per launch, what)
N26 = (1 << 26) - 4096 + 1
N20B = 256 * ((1 << 20) - 4096 + 1)
CASES = [
    ("c2", "fft_c1024_f32_dyn_kernel<0, 0, 1>", (1 << 20) * 16384, "C2 N=1024 cplx f32 fwd, batch 2^20"),
    ("c3", "TiledCfg<float, 13, 256, 3, 16, 32, 16", (1 << 16) * 131072, "C3 N=16384 real f32 fwd, batch 2^16"),
    ("c5", "TiledCfg<double, 10, 64, 3, 8, 16, 8", (1 << 20) * 32768, "C5 N=1024 cplx f64 fwd, batch 2^20"),
    ("c4_long", "fastconv_fused32_kernel", 8 * N26, "C4 FIR 2^26 samples, 4096 taps (8 B per output sample): 32-points-per-thread block kernel (fft_fir32.h, round 6)"),
    ("c4_batch", "fastconv_fused32_kernel", 8 * N20B, "C4 FIR 256 signals of 2^20 samples, 4096 taps: the same kernel"),
    ("c4_single", "fastconv_split1_kernel<", 8 * ((1 << 20) - 4095), "C4 stated call: 2^20 samples, 4096 taps, 255 reference-sized blocks, one-shot split kernel (fastconv_split1_kernel, round 4)"),
    ("fir_wave200", "fastconv_wave_kernel", 8 * ((1 << 26) - 199), "FIR 200 taps on 2^26 samples: one wavefront per 2048-sample block (round 3)"),
    ("conv1024", "fft_conv_kernel<pf::TiledCfg<float, 10, 64", (1 << 20) * 16384, "pffft_hip_convolve_batch N=1024 cplx f32, batch 2^20: forward x H backward in one kernel (round 4)"),
    ("stock3888", "SKP_f_3888_c_0", 2 * 34521 * 3888 * 8, "N=3888 cplx f32 forward unordered: Stockham plan 3 x 9 x 9 x 16 (round 4), 1 GiB of vectors"),
    ("big16_A", "tile_fft_kernel<float, 9, 8, 0, 1", 2 * (1 << 30), "N=2^16 cplx f32 = 512 x 128 (measured split, round 5), pass A (column tiles of 512 points), 1 GiB of vectors"),
    ("big16_Bi", "re:tile_fft_kernel<float, 7, 8, 0, 0, \\d, 1,", 2 * (1 << 30), "N=2^16 cplx f32, pass B (row tiles of 128 points) storing the internal layout"),
    ("blk_real", "big_block_kernel<float, 2>", 2 * (1 << 30), "real N=2^18 forward: pair pass + internal layout, one sweep"),
    ("stock4000", "SKP_f_4000_c_0", 2 * (1 << 15) * 4000 * 8, "N=4000 cplx f32 forward unordered (Stockham workgroup kernel), batch 2^15"),
    ("big20_A", "tile_fft_kernel<float, 10, 4, 0, 1", 2 * (1 << 30), "N=2^20 cplx f32, pass A"),
    ("big20_B", "tile_fft_kernel<float, 10, 4, 0, 0", 2 * (1 << 30), "N=2^20 cplx f32, pass B"),
    ("gen600k_A", "tileg_kernel<float, 1024, 0, 1", 2 * 223 * 600000 * 8, "N=600000 cplx f32 = 750 x 800, column pass on a run-time plan (fft_tileg.h, round 4)"),
    ("gen600k_B", "tileg_kernel<float, 1024, 0, 0", 2 * 223 * 600000 * 8, "N=600000 cplx f32, row pass on a run-time plan"),
    ("one12000", "fft_one_kernel<float, 0>", 2 * 11184 * 12000 * 8, "N=12000 cplx f32 forward ordered: single-image kernel 32 x 15 x 25 in place (fft_one.h, round 6), 1 GiB of vectors"),
    ("one24000r", "fft_one_kernel<float, 10>", 2 * 11184 * 24000 * 4, "real N=24000 f32 forward unordered on the single-image kernel: stages, pair pass in place, gather into the internal layout"),
    ("one9216d", "fft_one_kernel<double, 2>", 2 * 7281 * 9216 * 16, "N=9216 cplx f64 forward unordered on the single-image kernel: 12 x 8 x 8 x 12, last stage into the layout image"),
    ("c2_once12", "fft_c1024_f32_once_kernel<0, 0, 1, 4>", (1 << 12) * 16384, "N=1024 cplx f32 fwd, batch 2^12: one transform per wavefront in dispatch order (round 5)", 1024),
    ("c2_once14", "fft_c1024_f32_once_kernel<0, 0, 1, 4>", (1 << 14) * 16384, "N=1024 cplx f32 fwd, batch 2^14: one transform per wavefront in dispatch order (round 5)", 4096),
]
CASES = [c if len(c) == 5 else c + (None,) for c in CASES]
# c4_long / c4_batch share one kernel: told apart by the grid (both 256 workgroups) - by their position in the run instead:
# tools/prof_configs.py launches the long signal first, then the batch
vals = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for name in ("fetch", "write", "sq"):
    path = os.path.join(G, name, f"{name}_counter_collection.csv")
    rows = list(csv.DictReader(open(path)))
    for key, sub, _, _, wgs in CASES:
        for r in big_dispatches(rows, sub, "Kernel_Name", part=PART.get(key), wgs=wgs):
            vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[key] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size") if k in r}
            meta[key]["Kernel_Name"] = r["Kernel_Name"][:160]
import datetime, hashlib
def source_hash():   # the same hash bench.py stamps its line with: identifies the build the counters were captured on
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pffft_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]
out, traffic_json = {}, {"source": f"profiles/r{RND}_pmc.json", "source_hash": source_hash(),
                         "captured": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%MZ")}
lines = [f"# r{RND}: per-config PMC summary (tools/profile_round.sh, tools/prof_configs.py)\n",
         "HBM traffic = 2 x FETCH_SIZE[KiB] x 1024 + WRITE_SIZE[KiB] x 1024, the two counters from separate `--pmc` passes",
         "(MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream on gfx950).\n",
         "| config | kernel | avg ms (trace) | of 8 TB/s | HBM traffic / algorithmic | LDS conflict / active | VALU : LDS insts | wait-any / wave cycles | VGPR | scratch |",
         "|---|---|---|---|---|---|---|---|---|---|"]
for key, sub, alg, what, wgs in CASES:
    disp = big_dispatches(trace, sub, "Kernel_Name", "Grid_Size_X", part=PART.get(key), wgs=wgs)
    if not disp or key not in vals:
        continue
    disp = disp[-12:] if len(disp) > 24 else disp[1:]    # the first ~15 launches carry first-touch faults and the clock / TLB ramp
    durs = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in disp]
    krow = {"Name": disp[0]["Kernel_Name"], "Calls": len(disp), "AverageNs": sum(durs) / len(durs)}
    a = {k: sum(v) / len(v) for k, v in vals[key].items()}
    tr = 2 * a.get("FETCH_SIZE", 0) * 1024 + a.get("WRITE_SIZE", 0) * 1024
    ns = float(krow["AverageNs"])
    d = {"what": what, "kernel": krow["Name"][:200], "calls": int(krow["Calls"]), "avg_ns": ns, "algorithmic_bytes": alg,
         "frac_of_8TBps": alg / ns / 8000, "FETCH_SIZE_KiB": a.get("FETCH_SIZE"), "WRITE_SIZE_KiB": a.get("WRITE_SIZE"),
         "hbm_bytes_per_launch": tr, "traffic_over_algorithmic": tr / alg,
         "lds_conflict_fraction": a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_LDS_IDX_ACTIVE", 1), 1),
         "SQ_INSTS_VALU": a.get("SQ_INSTS_VALU"), "SQ_INSTS_LDS": a.get("SQ_INSTS_LDS"),
         "wait_any_over_wave_cycles": a.get("SQ_WAIT_ANY", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1),
         "valu_active_over_wave_cycles": a.get("SQ_ACTIVE_INST_VALU", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1),
         "dispatch": meta.get(key)}
    out[key] = d
    if key in ("c2", "c3", "c5", "c4_long"):
        traffic_json[key + "_bytes_per_launch"] = tr
    if key == "c4_long":            # bench.py: the trace-consistent fraction of the long FIR signal (back-to-back launches overlap)
        traffic_json["c4_long_kernel_ms_trace"] = ns / 1e6
    m = meta.get(key, {})
    lines.append(f"| {what} | `{krow['Name'][:70]}` | {ns/1e6:.4f} | {d['frac_of_8TBps']:.3f} | {d['traffic_over_algorithmic']:.3f} | "
                 f"{d['lds_conflict_fraction']:.3f} | {a.get('SQ_INSTS_VALU', 0):.3g} : {a.get('SQ_INSTS_LDS', 0):.3g} | "
                 f"{d['wait_any_over_wave_cycles']:.2f} | {m.get('VGPR_Count')} (+{m.get('Accum_VGPR_Count')} acc) | {m.get('Scratch_Size')} |")
json.dump(out, open(os.path.join(P, f"r{RND}_pmc.json"), "w"), indent=1)
json.dump(traffic_json, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
open(os.path.join(P, f"r{RND}_pmc.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
