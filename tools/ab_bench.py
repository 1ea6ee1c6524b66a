"""Interleaved A/B timing of kernel variants in ONE process (cdna guide §5.4 rule 24).
usage: python tools/ab_bench.py [variants...] [--batch-log2 20] [--rounds 5] [--mode fwd_u]"""
import argparse, os, sys, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="*", type=int, default=[0, 2])
ap.add_argument("--batch-log2", type=int, default=20)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--modes", default="fwd_u,fwd_o,bwd_u,bwd_o")
ap.add_argument("--check", action="store_true")
args = ap.parse_args()
B = 1 << args.batch_log2
s = pa.Setup(1024, pa.COMPLEX)
x = torch.rand(B, 2048, device="cuda") * 2 - 1
y = torch.empty_like(x)
modes = {"fwd_u": (pa.FORWARD, False), "fwd_o": (pa.FORWARD, True), "bwd_u": (pa.BACKWARD, False), "bwd_o": (pa.BACKWARD, True)}
res = {}
def timed(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
if args.check:
    pa.set_variant(2); ref = {m: s.transform_batch(x[:100003], None, *modes[m]).clone() for m in args.modes.split(",")}
    for v in args.variants:
        pa.set_variant(v)
        for m in args.modes.split(","):
            got = s.transform_batch(x[:100003], None, *modes[m])
            print("check variant", v, m, "bit-equal to variant 2:", bool(torch.equal(got, ref[m])))
for v in args.variants:
    pa.set_variant(v)
    for m in args.modes.split(","):
        s.transform_batch(x, y, *modes[m])
torch.cuda.synchronize()
for r in range(args.rounds):
    t = timed(lambda: y.copy_(x), args.reps); res.setdefault(("copy", "torch"), []).append(t)
    for m in args.modes.split(","):
        for v in args.variants:
            pa.set_variant(v)
            t = timed(lambda: s.transform_batch(x, y, *modes[m]), args.reps)
            res.setdefault((m, v), []).append(t)
pa.set_variant(0)
gb = B * 16384 / 1e6
for k, ts in res.items():
    med, best = statistics.median(ts), min(ts)
    print(f"{str(k):22s} median {med:7.3f} ms {gb/med:8.1f} GB/s | best {best:7.3f} ms {gb/best:8.1f} GB/s | {B/med/1e3:7.1f} M/s")
