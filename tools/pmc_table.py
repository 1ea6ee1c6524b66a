import csv, collections, sys
tag, pat = sys.argv[1], sys.argv[2]
for d in (f"prof_{tag}", f"prof_{tag}_2"):
    rows = list(csv.DictReader(open(f"gpurun_out/{d}/p_counter_collection.csv")))
    agg = collections.OrderedDict()
    for r in rows:
        k = r['Kernel_Name']
        if pat not in k: continue
        key = (r['Dispatch_Id'])
        a = agg.setdefault(key, {'name': k.split('(')[0].replace('void pf::', '')[:60], 'grid': r['Grid_Size'], 'wg': r['Workgroup_Size'], 'dur': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3})
        a[r['Counter_Name']] = float(r['Counter_Value'])
    seen = set()
    for k, a in agg.items():
        sig = (a['name'], a['grid'])
        if sig in seen: continue
        seen.add(sig)
        print({kk: (f"{vv:.4g}" if isinstance(vv, float) else vv) for kk, vv in a.items()})
