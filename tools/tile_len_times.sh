# Run on the GPU box (through gpurun): per-kernel average durations of every odd-stage tile length, both passes, float and double.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for dt in f32 f64; do
rm -rf gpurun_out/mrprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mrprof -o t -- python tools/tile_len_times.py $dt > gpurun_out/mrprof.log 2>&1
echo $dt
python - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/mrprof/**/t_kernel_stats.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r'tile_fft_kernel<(\w+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>', r['Name'])
    if m:
        T, logl, pp, d, seqc, pf, oi, ii, r0 = m.groups()
        rows.append((int(r0), int(logl), int(seqc), int(pf), int(r['Calls']), round(float(r['AverageNs'])/1e3, 1)))
for r in sorted(rows): print("R0=%d logl=%d L=%d %s pf=%d calls=%d %.1f us" % (r[0], r[1], r[0] << r[1], "A" if r[2] else "B", r[3], r[4], r[5]))
PY
done
tail -3 gpurun_out/mrprof.log
