"""Mean of every collected counter per (kernel, grid) of a rocprofv3 counter_collection.csv (development tool): python tools/r5_pmc_avg.py DIR [substring]"""
import collections, csv, glob, os, sys
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = collections.defaultdict(list)
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if sub in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:90], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, g, c), v in sorted(acc.items()):
    print(f"{c:24s} mean {sum(v) / len(v):14.1f}  n {len(v):4d}  grid {g:>9s}  {k}")
