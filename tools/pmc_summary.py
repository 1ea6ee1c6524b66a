"""Condense the rocprofv3 outputs of tools/profile.sh (gpurun_out/prof_*) into profiles/:
  profiles/<tag>_kernel_stats.csv    — `rocprofv3 --kernel-trace --stats` summary of the bench command
  profiles/<tag>_pmc.json / .md      — per-launch PMC numbers of the dominant kernel
  profiles/pmc_traffic.json          — what bench.py reports as roofline.traffic
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE come from SEPARATE --pmc passes,
both are in KiB, and on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream, so
the read side is doubled: traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024."""
import collections, csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
kern = sys.argv[2] if len(sys.argv) > 2 else "fft_c1024_f32_dyn_kernel<0, 0, 1>"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
shutil.copy(os.path.join(G, "prof_trace", "trace_kernel_stats.csv"), os.path.join(P, f"{tag}_kernel_stats.csv"))
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(G, "prof_trace", "trace_kernel_stats.csv")))}
krow = next(v for k, v in stats.items() if kern in k)
vals = collections.defaultdict(list)
meta = {}
for name in ("fetch", "write", "lds"):
    for r in csv.DictReader(open(os.path.join(G, f"prof_{name}", f"{name}_counter_collection.csv"))):
        if kern in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size")}
avg = {k: sum(v) / len(v) for k, v in vals.items()}
batch = 1 << 20
alg = batch * 16384
traffic = 2 * avg["FETCH_SIZE"] * 1024 + avg["WRITE_SIZE"] * 1024
out = {
    "kernel": krow["Name"], "calls_in_trace": int(krow["Calls"]), "avg_duration_ns": float(krow["AverageNs"]),
    "min_ns": float(krow["MinNs"]), "max_ns": float(krow["MaxNs"]),
    "batch": batch, "algorithmic_bytes_per_launch": alg,
    "achieved_GBps_from_trace": alg / float(krow["AverageNs"]),
    "FETCH_SIZE_KiB": avg["FETCH_SIZE"], "WRITE_SIZE_KiB": avg["WRITE_SIZE"],
    "hbm_bytes_per_launch": traffic, "traffic_over_algorithmic": traffic / alg,
    "SQ_LDS_BANK_CONFLICT": avg.get("SQ_LDS_BANK_CONFLICT"), "SQ_LDS_IDX_ACTIVE": avg.get("SQ_LDS_IDX_ACTIVE"),
    "lds_conflict_fraction": avg.get("SQ_LDS_BANK_CONFLICT", 0) / max(avg.get("SQ_LDS_IDX_ACTIVE", 1), 1),
    "SQ_INSTS_VALU_per_transform": avg.get("SQ_INSTS_VALU", 0) / batch,
    "SQ_INSTS_LDS_per_transform": avg.get("SQ_INSTS_LDS", 0) / batch,
    "SQ_WAVE_CYCLES": avg.get("SQ_WAVE_CYCLES"), "SQ_BUSY_CYCLES": avg.get("SQ_BUSY_CYCLES"),
    "dispatch": meta,
}
json.dump(out, open(os.path.join(P, f"{tag}_pmc.json"), "w"), indent=1)
json.dump({"c1024_fwd_unordered_bytes_per_launch": traffic, "source": f"profiles/{tag}_pmc.json"},
          open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
with open(os.path.join(P, f"{tag}_pmc.md"), "w") as f:
    f.write(f"# {tag}: rocprofv3 summary of `python bench.py --steps 5 --warmup 2` (tools/profile.sh)\n\n")
    f.write(f"Dominant kernel: `{krow['Name']}`\n\n")
    f.write("| quantity | value |\n|---|---|\n")
    f.write(f"| launches in trace / avg / min / max duration | {krow['Calls']} / {float(krow['AverageNs'])/1e6:.4f} ms / {float(krow['MinNs'])/1e6:.4f} / {float(krow['MaxNs'])/1e6:.4f} |\n")
    f.write(f"| algorithmic bytes per launch (2^20 transforms x 16 KiB) | {alg} |\n")
    f.write(f"| achieved (algorithmic bytes / avg duration) | {out['achieved_GBps_from_trace']:.1f} GB/s = {out['achieved_GBps_from_trace']/8000:.3f} of 8 TB/s |\n")
    f.write(f"| FETCH_SIZE (own pass, KiB) | {avg['FETCH_SIZE']:.0f} -> x2 x1024 = {2*avg['FETCH_SIZE']*1024:.4g} B read |\n")
    f.write(f"| WRITE_SIZE (own pass, KiB) | {avg['WRITE_SIZE']:.0f} -> x1024 = {avg['WRITE_SIZE']*1024:.4g} B written |\n")
    f.write(f"| HBM traffic / algorithmic | {traffic/alg:.4f} |\n")
    f.write(f"| SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE | {avg.get('SQ_LDS_BANK_CONFLICT',0):.4g} / {avg.get('SQ_LDS_IDX_ACTIVE',0):.4g} = {out['lds_conflict_fraction']:.3f} |\n")
    f.write(f"| VALU / LDS instructions per transform (wave-level) | {out['SQ_INSTS_VALU_per_transform']:.0f} / {out['SQ_INSTS_LDS_per_transform']:.0f} |\n")
    f.write(f"| dispatch | {meta} |\n")
print(open(os.path.join(P, f"{tag}_pmc.md")).read())
