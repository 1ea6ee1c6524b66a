"""Means of tools/size_scan.py outputs (gpurun_out/scans/ or profiles/r<NN>_scan_*): per file, complex / real, the mean over the four combinations,
the share of entries below 0.25 / 0.20, the sizes with a combination below 0.20, and the same restricted to the single-image sizes."""
import re, sys, statistics as st
for path in sys.argv[1:]:
    rows = []
    for l in open(path):
        m = re.match(r'(cplx|real) N=\s*(\d+) \[(\w+)\s*\] ([\d.]+) ([\d.]+) ([\d.]+) ([\d.]+)', l)
        if m: rows.append((m.group(1), int(m.group(2)), m.group(3), [float(m.group(i)) for i in range(4, 8)]))
    dbl = 'f64' in path
    for kind in ('cplx', 'real'):
        r = [x for x in rows if x[0] == kind]
        if not r: continue
        vals = [v for x in r for v in x[3]]
        one = [x for x in r if x[2] == 'fourstep' and 80000 < (x[1] if kind == 'cplx' else x[1] // 2) * (16 if dbl else 8) <= 147456]
        low = sorted([(x[1], min(x[3])) for x in r if min(x[3]) < 0.20], key=lambda t: t[1])
        print(f"{path.split('/')[-1]} {kind}: {len(r)} sizes, mean {st.mean(vals):.3f}, min {min(vals):.3f}, entries < 0.25: {sum(v < 0.25 for v in vals) / len(vals):.0%}, "
              f">= 0.70: {sum(v >= 0.70 for v in vals) / len(vals):.0%}, sizes with a combination < 0.20: {len(low)} {low[:6]}"
              + (f"; single-image sizes: {len(one)}, mean {st.mean(v for x in one for v in x[3]):.3f}" if one else ""))
