"""Condense gpurun_out/prof_mix* (tools/profile_mix.sh) into profiles/<tag>_mixers.md + <tag>_mixers_kernel_stats.csv.
HBM bytes as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE and WRITE_SIZE from separate passes, KiB units, read side
doubled on gfx950 (64 B counted per 128-B request of a wide coalesced stream)."""
import collections, csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
stats = [r for r in csv.DictReader(open(os.path.join(G, "prof_mix", "mix_kernel_stats.csv")))
         if "pfmix::" in r["Name"] or "fft_c1024" in r["Name"]]
with open(os.path.join(P, f"{tag}_mixers_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(stats[0].keys())); w.writeheader(); w.writerows(stats)
def counters(d):
    """per kernel-name substring: list of (start, {counter: value}, duration_ns) per dispatch, in launch order"""
    per = collections.OrderedDict()
    for r in csv.DictReader(open(os.path.join(G, d, "p_counter_collection.csv"))):
        e = per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "start": int(r["Start_Timestamp"]),
                                              "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "c": {}})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return sorted(per.values(), key=lambda e: e["start"])
fe, wr, sq = counters("prof_mix_fetch"), counters("prof_mix_write"), counters("prof_mix_sq")
trace = sorted(csv.DictReader(open(os.path.join(G, "prof_mix", "mix_kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
# (kernel substring, slice of its launches in program order, label, algorithmic bytes per launch)
want = [("mix_dyn_kernel<1, false>", slice(0, 12), "mixer, streaming kernel, 2^28 samples out of place + in place (1 lane)", 16 * (1 << 28)),
        ("mix_dyn_kernel<8, false>", slice(None), "mixer, streaming kernel, 2^28 samples (8 lanes: shift_recursive_osc_inp_c)", 16 * (1 << 28)),
        ("mix_dyn_kernel<1, false>", slice(12, None), "mixer as first pass of the two-pass shift+FFT (2^30 samples, variant 60)", 16 * (1 << 30)),
        ("fft_c1024_f32_mix_kernel<1>", slice(None), "shift fused into the N=1024 forward FFT, internal layout, batch 2^20", 16384 * (1 << 20)),
        ("fft_c1024_f32_mix_kernel<0>", slice(None), "shift fused into the N=1024 forward FFT, canonical, batch 2^20", 16384 * (1 << 20)),
        ("fft_c1024_f32_dyn_kernel<0, 0, 1>", slice(0, 6), "plain N=1024 forward FFT (same run)", 16384 * (1 << 20))]
md = [f"# {tag}: mixers (SURVEY.md §8 f-4) — rocprofv3 summary of `python tools/mix_bench.py` (tools/profile_mix.sh)\n",
      "Durations from the `--kernel-trace` pass, HBM bytes from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes "
      "(KiB; read side x2 on gfx950), SQ counters from a third pass; all per launch.\n",
      "| kernel | what | launches | avg ms (min) | algorithmic GB/s | of 8 TB/s | HBM bytes / algorithmic | VALU insts / KiB moved | wait-any / wave cycles |",
      "|---|---|---|---|---|---|---|---|---|"]
for sub, sl, label, alg in want:
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace if sub in r["Kernel_Name"]][sl]
    if not durs: continue
    def avg(lst, c):
        v = [e["c"][c] for e in [e for e in lst if sub in e["name"]][sl] if c in e["c"]]
        return sum(v) / len(v) if v else float("nan")
    traffic = 2 * avg(fe, "FETCH_SIZE") * 1024 + avg(wr, "WRITE_SIZE") * 1024
    ns = sum(durs) / len(durs)
    md.append(f"| `{sub}` | {label} | {len(durs)} | {ns/1e6:.3f} ({min(durs)/1e6:.3f}) | {alg/ns:.0f} | {alg/ns/8000:.3f} | "
              f"{traffic/alg:.4f} | {avg(sq,'SQ_INSTS_VALU')/(alg/1024):.1f} | {avg(sq,'SQ_WAIT_ANY')/max(avg(sq,'SQ_WAVE_CYCLES'),1):.2f} |")
md.append("\n`tools/mix_bench.py` output of the same build:\n\n```")
md += [l.rstrip() for l in open(os.path.join(G, "mix_bench.log")) if "amdgpu.ids" not in l]
md.append("```\n")
open(os.path.join(P, f"{tag}_mixers.md"), "w").write("\n".join(md))
print("\n".join(md))
