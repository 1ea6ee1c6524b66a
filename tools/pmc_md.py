"""gpurun_out/prof_<tag>{,_2} (tools/profile_cmd.sh) -> markdown table on stdout.  usage: pmc_md.py <tag> <kernel substring>"""
import csv, collections, re, sys
tag, pat = sys.argv[1], sys.argv[2]
agg = collections.OrderedDict()
for d in (f"prof_{tag}", f"prof_{tag}_2"):
    for r in csv.DictReader(open(f"gpurun_out/{d}/p_counter_collection.csv")):
        k = r["Kernel_Name"]
        if pat not in k: continue
        name = re.sub(r"void pf::|\(.*", "", k)
        key = (name, r["Grid_Size"], r["Workgroup_Size"])
        a = agg.setdefault(key, collections.defaultdict(list))
        a[r["Counter_Name"]].append(float(r["Counter_Value"]))
        a["dur_us" + d[-2:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        a["vgpr"] = [float(r["VGPR_Count"])]; a["lds"] = [float(r["LDS_Block_Size"])]; a["scratch"] = [float(r["Scratch_Size"])]
print("| kernel | grid / wg | VGPR | LDS B | scratch | us | VALU/LDS insts (M) | LDS conflict/active | wait-any / wave cycles | VALU active / wave cycles |")
print("|---|---|---|---|---|---|---|---|---|---|")
for (name, grid, wg), a in agg.items():
    m = {k: sum(v) / len(v) for k, v in a.items()}
    dur = min(v for k, v in m.items() if k.startswith("dur_us"))
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"| `{name[:70]}` | {grid} / {wg} | {m['vgpr']:.0f} | {m['lds']:.0f} | {m['scratch']:.0f} | {dur:.0f} | "
          f"{m.get('SQ_INSTS_VALU',0)/1e6:.1f} / {m.get('SQ_INSTS_LDS',0)/1e6:.1f} | "
          f"{m.get('SQ_LDS_BANK_CONFLICT',0)/max(m.get('SQ_LDS_IDX_ACTIVE',1),1):.3f} | {m.get('SQ_WAIT_ANY',0)/wc:.2f} | {m.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f} |")
