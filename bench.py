#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X pffft drop-in.

Metric (BASELINE.json): M transforms/s (+ GFLOPS + % of HBM roofline) on batched N=1024
complex-float forward FFTs — configs[1] ("N=1024 complex float fwd+inv, batch=1M, 1xMI355X").
A "step" is one pass of the hot path (pffft_transform semantics, internal-layout spectrum) over the
whole device-resident batch = ONE kernel launch through the C ABI (pffft_hip_transform_batch).

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5] [--scaling weak|strong]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU
over RCCL; it fails loudly when fewer than N GPUs are visible or when WORLD_SIZE disagrees with --gpus.

Multi-GPU: the batch shards across ranks with no data-path collective; RCCL only
combines the final number (all_reduce MAX of elapsed, SUM of transforms), bracketed by barriers.
  --scaling weak   (default) every rank transforms the config's per-GPU batch
  --scaling strong the config's TOTAL batch (c5: 2^23, BASELINE configs[4]) is split over the ranks and
                   transformed IN PLACE (in == out is legal, include/pffft/pffft.h:157): at 1 GPU 2^23
                   complex-double vectors are 128 GiB, out of place would need 256 GiB of the 288 GB

Prints ONE JSON line on rank 0.  `value` = whole-job throughput with inputs resident in HBM.
`roofline` prices the dominant kernel's ALGORITHMIC bytes (2 x vector bytes per transform,
8 B per FIR output sample) against 8 TB/s.  `cpu_baseline` times the real reference (oracle/_ref) on the
host cores of this box on a bounded sample of the same workload (test infrastructure, never the product).
At N = 1 the default line also carries `configs`: every other BASELINE config at its stated size, each with
its own `roofline` and `cpu_baseline` (no warm-up beyond --warmup; the first launches of each are reported as
`first_launches_ms` next to the timed region), and `sizes`: the reference's benchmark table
(benchmarks/bench_pffft.c:445,547-550,1140-1150: every size of its lists, real and complex, float and double, ordered and
unordered, forward and backward; plus eight legal sizes with factors 3 and 5 beyond LDS, which that list does not hold).  All GPU work runs back to back, the CPU baselines afterwards (an idle gap between
configs lets the clocks drop and the next config starts cold).  At N > 1 the line carries the C5 sharded config, weak and
strong, `ranks_seen` from the RCCL all-reduce and the fastest / slowest rank's ms_per_step.
Inputs are a counter hash of (seed, GLOBAL element index) (pffft_amd/sharding.py): a shard holds the same data whatever
the number of ranks, and the in-run parity check rebuilds its sampled vectors on the host from their indices.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12                            # MI355X HBM3E spec (MI355X_MICROARCH.md)
REAL, COMPLEX = 0, 1

# BASELINE.json configs; bytes/flops per unit from SURVEY.md §8(d) (5 N log2 N complex, 2.5 N log2 N real:
# benchmarks/bench_pffft.c:606)
CONFIGS = {
    "c2": dict(N=1024, tr=COMPLEX, dtype="f32", batch_log2=20, total_log2=23, bytes=16384, flops=51200,
               name="BASELINE configs[1]: N=1024 complex float forward"),
    "c3": dict(N=16384, tr=REAL, dtype="f32", batch_log2=16, total_log2=19, bytes=131072, flops=573440,
               name="BASELINE configs[2]: N=16384 real float forward"),
    "c5": dict(N=1024, tr=COMPLEX, dtype="f64", batch_log2=20, total_log2=23, bytes=32768, flops=51200,
               name="BASELINE configs[4]: N=1024 complex double forward"),
}
FIR = dict(signal_log2=20, taps=4096, long_log2=26,
           name="BASELINE configs[3]: pffastconv FIR, signal 2^20 real float, 4096 taps, overlap-save")


def _np_dtype(tag):
    return np.float64 if tag == "f64" else np.float32


def _host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def _cpubase():
    from oracle import ref as oref
    so = os.path.join(ROOT, "oracle", "_ref", "libcpubase.so")
    if not (oref.available() and os.path.exists(so)):
        return None, None
    lib = C.CDLL(so)
    lib.cpu_baseline_run.restype = C.c_double
    lib.cpu_baseline_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_long, C.c_int, C.c_int]
    if hasattr(lib, "cpu_baseline_fir"):
        lib.cpu_baseline_fir.restype = C.c_double
        lib.cpu_baseline_fir.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                         C.POINTER(C.c_int)]
    return lib, oref.get()


def cpu_baseline(cfg, sample_log2: int, target_s: float):
    """Time oracle/_ref (the reference's own SIMD path) on this host: loop of pffft_transform calls on the
    host threads that measure fastest, sharing one setup (include/pffft/pffft.h:102-105)."""
    lib, R = _cpubase()
    if lib is None:
        return None
    dt = _np_dtype(cfg["dtype"])
    a = R.api(dt)
    N, tr = cfg["N"], cfg["tr"]
    vs = N * (2 if tr == COMPLEX else 1)
    batch = 1 << sample_log2
    x = a.empty(batch * vs)
    x[:] = np.random.default_rng(2).uniform(-1, 1, x.size).astype(dt)
    y = a.empty(batch * vs)
    cores = _host_cores()
    isd = int(dt == np.float64)

    def run(threads, reps):
        return lib.cpu_baseline_run(N, tr, isd, 0, 0, x.ctypes.data, y.ctypes.data, batch, reps, threads)

    run(min(cores, 8), 1)  # warm-up (page faults, caches)
    t1 = run(1, 1)
    one_thread = batch / t1
    # the affinity mask can overstate what a container may use: pick the thread count that is
    # actually fastest on a short calibration, then spend the time budget there
    cands = sorted({c for c in (8, 16, 32, 64, 96, 128, 192, 256, cores) if c <= cores} | {min(cores, 4)})
    best_t, best_rate = 1, one_thread
    for th in cands:
        r = 2
        rate = r * batch / run(th, r)
        if rate > best_rate:
            best_t, best_rate = th, rate
    chunk = max(1, int(0.5 * best_rate / batch))
    reps, tall = 0, 0.0
    while tall < target_s and reps < 1000000:
        tall += run(best_t, chunk)
        reps += chunk
    rate = batch * reps / tall
    return {
        "value": round(rate / 1e6, 4), "unit": "M transforms/s", "cores": best_t, "kind": "reference",
        "sample": f"{reps} x 2^{sample_log2} transforms (N={N} {'cplx' if tr else 'real'} {cfg['dtype']} fwd, "
                  f"pffft{'d' if isd else ''}_transform loop, {tall:.1f} s wall, {best_t} threads sharing one setup)",
        "one_thread_value": round(one_thread / 1e6, 4), "simd_arch": a.simd_arch().decode(),
        "gflops": round(rate * cfg["flops"] / 1e9, 2),
    }


def cpu_baseline_fir(x: np.ndarray, h: np.ndarray, target_s: float):
    lib, R = _cpubase()
    if lib is None or not hasattr(lib, "cpu_baseline_fir"):
        return None
    a = R.f32
    xx, hh = a.aligned(x), a.aligned(h)
    cores = _host_cores()
    prod = C.c_int(0)

    def run(threads, reps):
        y = a.empty(threads * xx.size)
        return lib.cpu_baseline_fir(hh.ctypes.data, hh.size, xx.ctypes.data, xx.size, y.ctypes.data, reps, threads,
                                    C.byref(prod))

    t1 = run(1, 2) / 2
    n_out = prod.value
    best_t, best_rate, best_call = 1, n_out / t1, t1
    for th in sorted({c for c in (8, 16, 32, 64, 128, cores) if c <= cores}):
        tt = run(th, 2)
        rate = th * 2 * n_out / tt
        if rate > best_rate:
            best_t, best_rate, best_call = th, rate, tt / 2
    reps = max(2, min(100000, int(target_s / (1.5 * best_call))))   # the sustained rate is lower than the calibration's
    t = run(best_t, reps)
    rate = best_t * reps * n_out / t
    return {"value": round(rate / 1e9, 4), "unit": "G output samples/s", "cores": best_t, "kind": "reference",
            "sample": f"{best_t} threads x {reps} calls of pffastconv_apply on the 2^{FIR['signal_log2']}-sample signal "
                      f"({FIR['taps']} taps, own setup per thread, {t:.1f} s wall)",
            "one_thread_value": round(n_out / t1 / 1e9, 4), "one_call_ms_one_thread": round(t1 * 1e3, 3)}


# --------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """`--gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    if args.backend == "nccl" or not args.selftest_cpu:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus and os.environ.get("PFFFT_BENCH_SHARE_GPU") != "1":
            raise SystemExit(f"bench.py: --gpus {args.gpus} asked, {have} GPU(s) visible on this box — refusing to "
                             "report a multi-GPU number from fewer devices")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


class Timer:
    """HIP events on the launch stream (torch's current stream is the stream the C ABI launches on)."""

    def __init__(self, torch):
        self.torch = torch

    def __call__(self, fn, reps, warm=1):
        t = self.torch
        for _ in range(warm):
            fn()
        t.cuda.synchronize()
        a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        t.cuda.synchronize()
        return a.elapsed_time(b) * 1e-3 / reps


def source_hash():
    """sha256 over the kernel sources (pffft_amd/csrc/*.h, *.hip): identifies the build a PMC capture belongs to."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pffft_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def roofline(alg_bytes_per_launch, kernel_s, traffic_key=None):
    """`traffic` cannot be measured inside this run (PMC counters need rocprofv3 around the process): it is the figure of a
    `rocprofv3 --pmc` capture of the same workload (tools/profile_round.sh -> tools/pmc_round.py -> profiles/pmc_traffic.json) - replayed ONLY
    when that capture was taken on these very kernel sources (source hash) and is less than 24 h old; otherwise `traffic` is null and
    `traffic_source` says why (a replay that can drift from the build is not a measurement)."""
    ach = alg_bytes_per_launch / kernel_s
    out = {"bound": "hbm", "achieved": round(ach / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
           "frac": round(ach / HBM_PEAK, 4), "traffic": None, "kernel_ms": round(kernel_s * 1e3, 4),
           "algorithmic_bytes_per_launch": int(alg_bytes_per_launch)}
    if traffic_key:
        try:
            import datetime
            rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            v = rec.get(traffic_key)
            cap, when = rec.get("source_hash"), rec.get("captured", "")
            age_h = None
            try:
                t0 = datetime.datetime.strptime(when, "%Y-%m-%dT%H:%MZ").replace(tzinfo=datetime.timezone.utc)
                age_h = (datetime.datetime.now(datetime.timezone.utc) - t0).total_seconds() / 3600.0
            except Exception:
                pass
            same = cap == source_hash()
            fresh = age_h is not None and -6.0 <= age_h <= 24.0      # (clock skew between the capturing and the running box)
            if v is not None and same and fresh:
                out["traffic"] = int(round(v))
                out["traffic_source"] = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes, 2 x FETCH_SIZE + "
                                         f"WRITE_SIZE, MI355X_MICROARCH.md HBM section) captured {when} ({age_h:.1f} h ago) on source hash {cap} = this build; "
                                         "not measured inside this run")
            elif v is not None:
                out["traffic_source"] = (f"null: the capture in profiles/pmc_traffic.json ({when}, source hash {cap}) is "
                                         + ("of another build" if not same else "older than 24 h") + f" (this build: {source_hash()}); "
                                         "re-capture with tools/profile_round.sh + tools/pmc_round.py")
        except Exception:
            pass
    return out


def make_input(torch, dev, batch, vec_scalars, tdt, seed, first_vector=0):
    """Uniform [-1, 1): counter hash of (seed, GLOBAL element index), generated on the device (pffft_amd/sharding.py).
    `first_vector` = global index of this shard's first vector: the same vectors whatever the number of ranks."""
    from pffft_amd.sharding import global_uniform_
    x = torch.empty(batch, vec_scalars, device=dev, dtype=tdt)
    return global_uniform_(x, first_vector * vec_scalars, seed)


def make_input_into(torch, x, seed, first_vector=0):
    from pffft_amd.sharding import global_uniform_
    return global_uniform_(x, first_vector * x.shape[1], seed)


def sample_indices(batch, want=4096):
    """>= `want` transforms of a batch: both ends, the middle, and a stride over everything."""
    step = max(1, batch // want)
    return sorted({0, 1, batch // 2, batch - 2, batch - 1} & set(range(batch)) | set(range(3, batch, step)))


def fft_config_run(torch, pa, dev, cfg, batch, first_vector, steps, warmup, in_place, dist, rank, seed):
    """One FFT config on this rank's shard: W warm-up + K timed steps inside the barrier bracket.
    Returns (elapsed_s of this rank, kernel_s from HIP events, kernel name, parity, first launches in ms)."""
    from pffft_amd.sharding import global_uniform_np, timed_steps
    dt = _np_dtype(cfg["dtype"])
    tdt = torch.float64 if cfg["dtype"] == "f64" else torch.float32
    setup = pa.Setup(cfg["N"], cfg["tr"], dt)
    vs = setup.vec_scalars
    x = make_input(torch, dev, batch, vs, tdt, seed, first_vector)
    y = x if in_place else torch.empty_like(x)

    def step():
        setup.transform_batch(x, y, pa.FORWARD, ordered=False)

    # the first launches of this kernel on these buffers, timed one by one (cold: clocks, TLBs).  They ARE the first warm-up
    # steps: the untimed launches before the timed region add up to exactly `warmup`.
    ncold = min(6, warmup)
    first = []
    for _ in range(ncold):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); step(); b.record()
        first.append((a, b))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    count = [0]
    rest = warmup - ncold

    def timed_step():
        if count[0] == rest:
            e0.record()
        step()
        count[0] += 1
        if count[0] == rest + steps:
            e1.record()

    elapsed = timed_steps(timed_step, steps, rest, dist, torch.cuda.synchronize)
    kernel_s = e0.elapsed_time(e1) * 1e-3 / steps
    first_ms = [round(a.elapsed_time(b), 4) for a, b in first]
    # parity against the real reference when it travelled, AFTER the timed region (host work before it would leave the GPU
    # idle and the timed region cold): >= 4096 sampled transforms whose inputs are rebuilt on the host from their GLOBAL
    # indices.  In place: the buffer is regenerated and transformed once more.  (tests/ are the gate.)
    parity = None
    idx = sample_indices(batch)
    try:
        from oracle import ref as oref
        if oref.available():
            if in_place:
                make_input_into(torch, x, seed, first_vector)
                step()
            torch.cuda.synchronize()
            rs = oref.get().setup(cfg["N"], cfg["tr"], dt)
            xin = np.stack([global_uniform_np((first_vector + i) * vs, vs, seed, dt) for i in idx])
            want = rs.batch(xin, oref.FORWARD, False)
            got = y[torch.tensor(idx, device=dev)].cpu().numpy()
            num = np.abs(got.astype(np.float64) - want).max(axis=1)
            parity = {"max_rel_err": float((num / np.abs(want).max(axis=1)).max()), "transforms_checked": len(idx)}
            rs.close()
    except Exception as e:
        parity = f"unchecked: {e}"
    kname = pa.kernel_name(setup)
    setup.close()
    del x, y
    torch.cuda.empty_cache()
    return elapsed, kernel_s, kname, parity, first_ms


# the reference's benchmark lists (benchmarks/bench_pffft.c:1140-1150 with powers of two, :1153-1171)
REF_SIZES = [64, 96, 128, 160, 192, 256, 384, 480, 512, 640, 768, 800, 1024, 2048, 2400, 4096, 8192, 9216, 16384, 32768,
             262144, 1048576]


# Sizes with factors 3 and 5 beyond LDS (legal sizes of tests/test_fft_factors.c the reference's benchmark list does not hold):
# two tile passes (15360 = 64 x 240, 61440 = 256 x 240, 102400 = 400 x 256, 368640 = 480 x 768: pffft_hip_tile_plan), three where the
# streaming route needs five sweeps (1024000 = 64 x 200 x 80), one size with 2^5 only (12000 = 100 x 120 float complex on run-time plans, three streaming passes otherwise), and -
# round 4 - two sizes on the run-time tile lengths of fft_tileg.h (384000 = 480 x 800, 600000 = 750 x 800; real N: half of it) - DESIGN.md §3.5
BEYOND_LDS_35 = [12000, 15360, 61440, 102400, 368640, 384000, 600000, 1024000]


def sizes_table(torch, pa, dev, timer, R=None):
    """The reference's benchmark table (benchmarks/bench_pffft.c:445,547-550: every size of its lists, real and complex,
    "PFFFT" = ordered and "PFFFT-U" = unordered, forward and backward - the reference times the pair; both halves are listed
    here) in float and double: 1 GiB of vectors per launch, 60 untimed launches per size, then 10 untimed + 20 timed launches per combination, fraction of 8 TB/s on
    2 x vector bytes per transform (a 256 MiB launch lasts ~80 us: start-up, tail and the gap to the next launch cost 15-20 %).  One line per kernel family and layout, so that a regression shows up in the driver's
    record without profiles/."""
    out = {"workload": "1 GiB of vectors per launch, best of two runs of 10 + 20 launches; [fwd ordered, fwd unordered, bwd ordered, bwd unordered] "
                       "as fractions of 8 TB/s on 2 x vector bytes; after timing the last vector of every launch is checked against "
                       "oracle/_ref (parity_worst_rel_err: per precision, over all sizes and combinations)",
           "sizes": REF_SIZES, "sizes_with_factors_3_5_beyond_lds": BEYOND_LDS_35}
    worst = {}
    for tag, dt, tdt in (("f32", np.float32, torch.float32), ("f64", np.float64, torch.float64)):
        isz = np.dtype(dt).itemsize
        pool = make_input(torch, dev, 1, (1 << 30) // isz, tdt, seed=7).reshape(-1)
        ypool = torch.empty_like(pool)
        for tr, name in ((pa.COMPLEX, "complex"), (pa.REAL, "real")):
            tab = {}
            beyond = []
            both = []
            for N in REF_SIZES + BEYOND_LDS_35:
                s = pa.Setup(N, tr, dt)
                if pa.kernel_name(s) == "fourstep":
                    beyond.append(N)
                batch = max(1, pool.numel() // s.vec_scalars)
                x = pool[: batch * s.vec_scalars].view(batch, s.vec_scalars)
                y = ypool[: batch * s.vec_scalars].view(batch, s.vec_scalars)
                row = []
                # a new setup uploads its tables with synchronous copies: the GPU idles for a moment and the clocks take
                # ~30 ms of work to come back (first_launches_ms of the configs shows the same ramp) - without this untimed
                # run the FIRST of the four combinations of every size reads 0.05-0.10 low
                timer(lambda: s.transform_batch(x, y, pa.FORWARD, ordered=True), 1, warm=60)
                rs = R.setup(N, tr, dt) if R is not None else None
                xl = x[batch - 1].cpu().numpy() if rs is not None else None
                for d in (pa.FORWARD, pa.BACKWARD):
                    for o in (True, False):
                        # best of two runs of 10 + 20 launches: single runs of the small-vector kernels scatter by 0.05-0.08 from one
                        # run to the next on the same build (r4_ab.py (earlier-round tool, git history)), which read as regressions that were not there
                        # (a best-of-two figure is not comparable with the single runs of earlier rounds - the mean of the two
                        #  runs is summed up beside it, `mean_of_two_runs` in the summary)
                        t2 = [timer(lambda: s.transform_batch(x, y, d, ordered=o), 20, warm=10) for _ in range(2)]
                        t = min(t2)
                        row.append(round(2 * x.numel() * isz / t / HBM_PEAK, 3))
                        both.append((N, 2 * x.numel() * isz / (0.5 * (t2[0] + t2[1])) / HBM_PEAK))
                        if rs is not None:      # the timed launches left the spectrum of the last vector in y
                            want = (rs.transform_ordered if o else rs.transform_unordered)(xl, d)
                            got = y[batch - 1].cpu().numpy().astype(np.float64)
                            e = float(np.abs(got - want).max() / np.abs(want).max())
                            # (double with factors 3 / 5: the reference's float-suffixed constants make IT the inexact side)
                            key = tag if (dt == np.float32 or N & (N - 1) == 0) else tag + "_factors_3_5_vs_inexact_reference"
                            if e > worst.get(key, (0.0,))[0]:
                                worst[key] = (e, N, name, "fwd" if d == pa.FORWARD else "bwd", "ordered" if o else "unordered")
                tab[str(N)] = row
                s.close()
                if rs is not None:
                    rs.close()
            out[f"{tag}_{name}"] = tab
            out[f"{tag}_{name}_beyond_lds"] = beyond
            far = set(beyond)
            out.setdefault("mean_of_two_runs", {})[f"{tag}_{name}"] = {
                "lds_resident_mean": round(float(np.mean([v for n, v in both if n not in far])), 3),
                "beyond_lds_mean": round(float(np.mean([v for n, v in both if n in far])), 3)}
        del pool, ypool
        torch.cuda.empty_cache()
    out["parity_worst_rel_err"] = {k: {"err": float("%.3g" % v[0]), "at": list(v[1:])} for k, v in worst.items()} if worst else "unchecked: oracle/_ref not present"
    return out


def sizes_summary(tab):
    """[min, mean, share >= 0.70] of the LDS-resident entries (N <= 16384) and [min, mean] beyond, per precision / transform."""
    res = {}
    for key, t in tab.items():
        if not isinstance(t, dict) or not key.startswith("f") or key.endswith("_beyond_lds"):
            continue
        far = set(tab.get(key + "_beyond_lds", []))          # sizes on the tile / streaming passes ("fourstep")
        inl = [v for n, row in t.items() if int(n) not in far for v in row]
        big = [v for n, row in t.items() if int(n) in far for v in row]
        res[key] = {"lds_resident": [min(inl), round(float(np.mean(inl)), 3), round(float(np.mean([v >= 0.70 for v in inl])), 2)],
                    "beyond_lds": [min(big), round(float(np.mean(big)), 3)]}
    return res


def batch_sweep(torch, pa, dev, timer):
    """The batch axis (the reference's bench times every size for >= 150 ms per point: benchmarks/bench_pffft.c:547-550,1004-1017): C2 and
    C5 at batch 2^10 .. 2^20 in powers of 4, forward unordered, out of place, back-to-back launches through the C ABI - us per launch
    (HIP events around the run: below ~8 us per launch the runtime's launch path binds, not the kernel) and the fraction of 8 TB/s on
    the algorithmic bytes.  Best of three runs of >= 20 ms each; the launch shape the planner picks per batch is pffft_hip_describe()'s."""
    out = {"protocol": "forward unordered, out of place, back-to-back launches, best of 3 runs of >= 20 ms; [us per launch, fraction of 8 TB/s]"}
    for key in ("c2", "c5"):
        cfg = CONFIGS[key]
        dt = _np_dtype(cfg["dtype"])
        tdt = torch.float64 if cfg["dtype"] == "f64" else torch.float32
        s = pa.Setup(cfg["N"], cfg["tr"], dt)
        x = make_input(torch, dev, 1 << 20, s.vec_scalars, tdt, seed=11)
        y = torch.empty_like(x)
        row = {}
        for lg in range(10, 21, 2):
            b = 1 << lg
            xb, yb = x[:b], y[:b]
            est = max(8e-6, b * cfg["bytes"] / (0.5 * HBM_PEAK))
            reps = int(min(4000, max(10, 0.02 / est)))
            t = min(timer(lambda: s.transform_batch(xb, yb, pa.FORWARD, ordered=False), reps, warm=5) for _ in range(3))
            row[f"2^{lg}"] = [round(t * 1e6, 2), round(b * cfg["bytes"] / t / HBM_PEAK, 3)]
        out[key] = row
        try:
            out[key + "_routes"] = pa.describe(s).strip().split("\n")[2].strip()
        except Exception:
            pass
        s.close()
        del x, y
        torch.cuda.empty_cache()
    return out


def conv_config(torch, pa, dev, timer, warmup):
    """pffft_hip_convolve_batch on the C2 shape: forward x H backward of 2^20 vectors of N = 1024 complex float, one filter
    spectrum; roofline = one read + one write of every vector."""
    N, b = 1024, 1 << 20
    s = pa.Setup(N, COMPLEX, np.float32)
    x = make_input(torch, dev, b, 2 * N, torch.float32, seed=8)
    y = torch.empty_like(x)
    H = s.transform_batch(make_input(torch, dev, 1, 2 * N, torch.float32, seed=9), None, pa.FORWARD, ordered=False).reshape(-1).contiguous()
    t = timer(lambda: s.convolve_batch(x, H, out=y, scaling=1.0 / N), 20, warm=max(3, warmup))
    rec = {"workload": "pffft_hip_convolve_batch, N=1024 complex float, batch 2^20, broadcast filter spectrum (fused kernel)",
           "value": round(b / t / 1e6, 2), "unit": "M convolutions/s", "roofline": roofline(b * 16384, t)}
    pa.set_variant(120)
    try:
        tc = timer(lambda: s.convolve_batch(x, H, out=y, scaling=1.0 / N), 10, warm=3)
    finally:
        pa.set_variant(0)
    rec["composed_three_launches_Mps"] = round(b / tc / 1e6, 2)
    try:
        from oracle import ref as oref
        if oref.available():
            rs = oref.get().setup(N, COMPLEX, np.float32)
            s.convolve_batch(x, H, out=y, scaling=1.0 / N)
            idx = [0, 1, b // 2, b - 1]
            Hh = H.cpu().numpy()
            worst = 0.0
            for i in idx:
                X = rs.transform_unordered(x[i].cpu().numpy(), oref.FORWARD)
                Y = rs.zconvolve(X, Hh, np.zeros_like(X), 1.0 / N, accumulate=False)
                w = rs.transform_unordered(Y, oref.BACKWARD)
                worst = max(worst, float(np.abs(y[i].cpu().numpy() - w).max() / np.abs(w).max()))
            rec["parity_max_rel_err"] = worst
            rs.close()
    except Exception as e:
        rec["parity_max_rel_err"] = f"unchecked: {e}"
    s.close()
    del x, y
    torch.cuda.empty_cache()
    return rec


VALU_PEAK = 108.0 * 256 * 2.4e9   # float results / s: 108 per clock and CU measured (profiles/r02_probes.md), 256 CUs, 2.4 GHz
VALU_PEAK_SPEC = 157.3e12         # vector FP32 peak of the data sheet (MI355X_MICROARCH.md: 128 lanes x FMA x 256 CUs x 2.4 GHz)


def fir_flops_per_output(nfft, taps):
    """per block 2 x (2.5 Nfft log2 Nfft) + 6 (Nfft / 2) + Nfft flops, Nfft - taps + 1 outputs."""
    return (5.0 * nfft * np.log2(nfft) + 3.0 * nfft + nfft) / (nfft - taps + 1)


def fir_config(torch, pa, dev, timer, warmup):
    """BASELINE configs[3].  The single 2^20-sample call is latency-bound (255 blocks); the throughput regime is
    measured on (i) a batch of independent 2^20-sample signals through pffastconv_hip_apply_batch and (ii) one
    2^26-sample signal — `roofline` is (ii): 8 B per output sample (read x once, write y once), with a second leg
    `roofline_valu` that prices the same run against the measured VALU peak (at 4096 taps the arithmetic binds before HBM).
    Returns (record, closure that adds the CPU baseline later)."""
    taps, L = FIR["taps"], 1 << FIR["signal_log2"]
    h = np.random.default_rng(4).uniform(-1, 1, taps).astype(np.float32)
    out = {"workload": FIR["name"]}
    fc = pa.FastConv(h, 0, 0)
    sig = make_input(torch, dev, 1, L, torch.float32, seed=4).reshape(-1)
    y = torch.empty_like(sig)
    t = timer(lambda: fc.apply(sig, True, out=y), 200, warm=max(3, warmup))
    n_out = L - taps + 1
    out["single_call_us"] = round(t * 1e6, 2)
    out["single_call_Gsamples_per_s"] = round(n_out / t / 1e9, 2)
    out["single_call_frac"] = round(8 * n_out / t / HBM_PEAK, 4)
    # parity of the same call against the real reference (spot check; tests/ are the gate)
    x_host = sig.cpu().numpy()
    try:
        from oracle import ref as oref
        if oref.available():
            yw, nw, _ = oref.get().fastconv(x_host, h, 0, 0, 1)
            got = y[:nw].cpu().numpy()
            out["parity_max_err_over_range"] = float(np.abs(got - yw).max() / (yw.max() - yw.min()))
    except Exception as e:
        out["parity_max_err_over_range"] = f"unchecked: {e}"
    if hasattr(fc, "apply_batch"):
        nsig = 256
        xs = make_input(torch, dev, nsig, L, torch.float32, seed=5)
        ys = torch.empty_like(xs)
        t = timer(lambda: fc.apply_batch(xs, True, out=ys), 20, warm=warmup)
        out["batch_signals"] = nsig
        out["batch_Gsamples_per_s"] = round(nsig * n_out / t / 1e9, 2)
        out["batch_frac"] = round(8 * nsig * n_out / t / HBM_PEAK, 4)
        del xs, ys
    Ll = 1 << FIR["long_log2"]
    xl = make_input(torch, dev, 1, Ll, torch.float32, seed=6).reshape(-1)
    yl = torch.empty_like(xl)
    first = []
    for _ in range(6):
        first.append(round(timer(lambda: fc.apply(xl, True, out=yl), 1, warm=0) * 1e3, 4))
    t = timer(lambda: fc.apply(xl, True, out=yl), 40, warm=warmup)
    nl = Ll - taps + 1
    out["value"] = round(nl / t / 1e9, 2)
    out["unit"] = "G output samples/s"
    out["long_signal"] = f"2^{FIR['long_log2']} samples, {taps} taps"
    out["first_launches_ms"] = first
    out["roofline"] = roofline(8 * nl, t, "c4_long_bytes_per_launch")
    # Back-to-back launches of this persistent kernel overlap: the next launch's workgroups start on the CUs the tail of this one has left,
    # so the step time measured by HIP events is SHORTER than the kernel's own duration in a rocprofv3 trace (round 4: 0.2013 against 0.2269 ms).
    # The contract's fraction is algorithmic bytes over the kernel's duration: where a trace of this build's kernel exists
    # (profiles/pmc_traffic.json, the capture that also supplies `traffic`) `frac` is the trace-consistent one and the event figure is kept
    # beside it as `frac_events`.
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        tr_ms = rec.get("c4_long_kernel_ms_trace")
        rf = out["roofline"]
        if tr_ms and rec.get("source_hash") == source_hash():
            rf["frac_events"], rf["kernel_ms_events"] = rf["frac"], rf["kernel_ms"]
            rf["kernel_ms_trace"] = round(float(tr_ms), 4)
            rf["frac"] = round(min(rf["frac"], 8 * nl / (float(tr_ms) * 1e-3) / HBM_PEAK), 4)
            rf["achieved"] = round(rf["frac"] * HBM_PEAK / 1e9, 1)
            rf["frac_note"] = ("frac = min(HIP-event step time, rocprofv3 kernel duration of the capture of THIS build) - overlapping "
                               f"back-to-back launches make the step time the shorter one; source hash {source_hash()}")
        elif tr_ms:
            # (a trace of another build must not pin the figure - it would hide a real change of the kernel)
            rf["frac_note"] = (f"frac from HIP events; the rocprofv3 trace in profiles/pmc_traffic.json ({float(tr_ms):.4f} ms) belongs to source hash "
                               f"{rec.get('source_hash')}, this build is {source_hash()}: stale, not used")
    except Exception:
        pass
    nfft_used = 16384 if taps >= 1024 else 8192    # internal block of the throughput regime (pffastconv_impl.h fc_big_nfft)
    fl = fir_flops_per_output(nfft_used, taps) * nl
    out["roofline_valu"] = {"bound": "valu", "achieved": round(fl / t / 1e12, 2), "peak": round(VALU_PEAK / 1e12, 1), "unit": "TFLOP/s",
                            "frac": round(fl / t / VALU_PEAK, 4), "peak_spec": round(VALU_PEAK_SPEC / 1e12, 1),
                            "frac_spec": round(fl / t / VALU_PEAK_SPEC, 4), "flops_per_output_sample": round(fl / nl, 1),
                            "note": f"SURVEY.md 8(d) C4 flop count at the internal block length {nfft_used}; peak = 108 float results "
                                    "per clock and CU measured (profiles/r02_probes.md) x 256 CUs x 2.4 GHz, one flop per result; "
                                    "peak_spec = the data sheet's 157.3 TFLOP/s"}
    # LDS roof from the block kernel's own structure (round 6, fft_fir32.h: FOUR exchanges of the 64 KiB image per 16384-sample block, each
    # one store sweep and one load sweep; round 5's split kernel: 448 KiB of stores + 512 KiB of loads) at the guide's rates
    # (MI355X_MICROARCH.md, LDS table: ds_write_b128 ~79 B/clk and CU, ds_read2_b64 128 B/clk and CU), all 256 CUs at 2.4 GHz
    if nfft_used == 16384:
        step_b = nfft_used - taps + 1
        nblk = (nl + step_b - 1) // step_b
        st_b, ld_b = 4 * 65536, 4 * 65536
        t_lds = nblk * (st_b / 79.0 + ld_b / 128.0) / (256 * 2.4e9)
        out["roofline_lds"] = {"bound": "lds", "store_bytes_per_block": st_b, "load_bytes_per_block": ld_b, "blocks": int(nblk),
                               "floor_ms": round(t_lds * 1e3, 4), "frac": round(t_lds / t, 4),
                               "note": "fraction of the step time the LDS pipe alone needs: exchanges per block x 64 KiB each way at 79 (stores) / "
                                       "128 (loads) B per clock and CU, 256 CUs, 2.4 GHz"}
        roofs = {"hbm": out["roofline"]["frac"], "valu": out["roofline_valu"]["frac"], "lds": out["roofline_lds"]["frac"]}
        binding = max(roofs, key=roofs.get)
        out["roofline"]["bound"] = binding
        out["roofline"]["bound_note"] = ("the roof that binds this kernel (largest of the three fractions " + json.dumps(roofs) + "): `achieved` / `peak` / `frac` "
                                         "of this dict stay the HBM figures of the contract (8 B per output sample); the binding roof's own figures are in "
                                         f"roofline_{binding}" if binding != "hbm" else "HBM binds")
    fc.close()
    del xl, yl, sig, y
    torch.cuda.empty_cache()

    def add_cpu(cpu_seconds):
        cb = cpu_baseline_fir(x_host, h, cpu_seconds) if cpu_seconds > 0 else None
        out["cpu_baseline"] = cb if cb else {"value": None, "unit": "G output samples/s", "cores": 0, "kind": "reference",
                                             "sample": "oracle/_ref not present on this box"}
    return out, add_cpu


def fake_main(args, world, rank):
    """Harness self-test on CPU (tests/test_dist.py): same spawn path, gloo, a host stand-in step — the transform
    itself has no CPU implementation in the product."""
    import torch
    import torch.distributed as dist
    from pffft_amd.sharding import combine_stats, timed_steps
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    units = 1000.0
    x = np.random.default_rng(rank).standard_normal((64, 64))
    el = timed_steps(lambda: np.fft.fft(x, axis=1), args.steps, args.warmup, dist if world > 1 else None, None)
    st = combine_stats(el, units, dist if world > 1 else None, torch.device("cpu"))
    el, tot = st["elapsed_max"], st["units"]
    if rank == 0:
        print(json.dumps({"metric": "harness self-test (no transform ran)", "value": tot * args.steps / el, "unit": "units/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "selftest": True,
                          "ranks_seen": st["ranks_seen"],
                          "ms_per_step_fastest_rank": round(st["elapsed_min"] / args.steps * 1e3, 4),
                          "ms_per_step_slowest_rank": round(st["elapsed_max"] / args.steps * 1e3, 4)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def single_process_main(args):
    """--single-process: ONE host process (one rank, one thread) drives --gpus N devices through pffft[d]_hip_transform_batch_multi with
    ONE shared setup (round 6: a setup keeps its device state per device, include/pffft_hip.h) - the C-level way to shard a batch over a
    node, no torch.distributed, no RCCL: the path has no collective.  Same contract: W untimed + K timed
    steps, a step = one _multi call (every device transforms its shard of the config's per-GPU batch, weak scaling), the timed region
    bracketed by a synchronisation of ALL devices on both sides, value = whole-job transforms per second.  PFFFT_BENCH_SHARE_GPU=1 (harness
    test on a 1-GPU box, never a reportable number): the parts share the visible devices round robin."""
    import ctypes as C
    import time
    import torch
    import pffft_amd as pa
    from pffft_amd.sharding import global_uniform_np
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.config not in ("c2", "c3", "c5") or args.scaling != "weak":
        raise SystemExit("bench.py --single-process: c2 / c3 / c5, weak scaling")
    share = os.environ.get("PFFFT_BENCH_SHARE_GPU") == "1"
    nvis = torch.cuda.device_count()
    if nvis < args.gpus and not share:
        raise SystemExit(f"bench.py --single-process: --gpus {args.gpus} but {nvis} device(s) visible")
    cfg = CONFIGS[args.config]
    seed = {"c2": 2, "c3": 3, "c5": 5}[args.config]
    dt = _np_dtype(cfg["dtype"])
    tdt = torch.float64 if cfg["dtype"] == "f64" else torch.float32
    batch = 1 << (args.batch_log2 if args.batch_log2 is not None else cfg["batch_log2"])
    setup = pa.Setup(cfg["N"], cfg["tr"], dt)
    vs = setup.vec_scalars
    P = args.gpus
    devs = [p % nvis for p in range(P)]
    xs, ys, sts = [], [], []
    for p, d in enumerate(devs):
        torch.cuda.set_device(d)
        dev = torch.device("cuda", d)
        xs.append(make_input(torch, dev, batch, vs, tdt, seed, p * batch))     # part p holds global vectors [p batch, (p + 1) batch)
        ys.append(torch.empty_like(xs[-1]))
        sts.append(torch.cuda.Stream(device=dev))
    torch.cuda.set_device(devs[0])
    L = pa.lib()
    multi = L.pffftd_hip_transform_batch_multi if cfg["dtype"] == "f64" else L.pffft_hip_transform_batch_multi
    multi.restype = C.c_int
    a_dev = (C.c_int * P)(*devs)
    a_set = (C.c_void_p * P)(*([setup.handle] * P))
    a_in = (C.c_void_p * P)(*[t.data_ptr() for t in xs])
    a_out = (C.c_void_p * P)(*[t.data_ptr() for t in ys])
    a_b = (C.c_size_t * P)(*([batch] * P))
    a_st = (C.c_void_p * P)(*[q.cuda_stream for q in sts])

    def step():
        rc = multi(P, a_dev, a_set, a_in, a_out, a_b, pa.FORWARD, 0, a_st)
        if rc:
            raise SystemExit("pffft_hip_transform_batch_multi failed: " + pa.last_error())

    def sync_all():
        for d in sorted(set(devs)):
            torch.cuda.synchronize(d)

    for _ in range(args.warmup):
        step()
    sync_all()
    ev = []
    for p, d in enumerate(devs):                                   # per-device kernel time: HIP events on the part's own stream
        torch.cuda.set_device(d)
        ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
        ev[-1][0].record(sts[p])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    for p, d in enumerate(devs):
        torch.cuda.set_device(d)
        ev[p][1].record(sts[p])
    sync_all()
    el = time.perf_counter() - t0
    torch.cuda.set_device(devs[0])
    per_dev_ms = [a.elapsed_time(b) / args.steps for a, b in ev]
    tps = P * batch * args.steps / el
    # parity of the first and the last part against the reference, inputs rebuilt on the host from their GLOBAL indices
    parity = None
    try:
        from oracle import ref as oref
        if oref.available():
            rs = oref.get().setup(cfg["N"], cfg["tr"], dt)
            worst, cnt = 0.0, 0
            for p in sorted({0, P - 1}):
                idx = sample_indices(batch, 512)
                xin = np.stack([global_uniform_np((p * batch + i) * vs, vs, seed, dt) for i in idx])
                want = rs.batch(xin, oref.FORWARD, False)
                got = ys[p][torch.tensor(idx, device=ys[p].device)].cpu().numpy()
                worst = max(worst, float((np.abs(got.astype(np.float64) - want).max(axis=1) / np.abs(want).max(axis=1)).max()))
                cnt += len(idx)
            parity = {"max_rel_err": worst, "transforms_checked": cnt}
            rs.close()
    except Exception as e:
        parity = f"unchecked: {e}"
    seen = pa.setup_devices(setup)
    out = {
        "metric": "M transforms/s, batched N=%d %s-%s forward FFT (pffft_transform semantics)" % (
            cfg["N"], "complex" if cfg["tr"] else "real", "float" if cfg["dtype"] == "f32" else "double"),
        "value": round(tps / 1e6, 3), "unit": "M transforms/s", "n_gpus": P, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": cfg["dtype"],
        "data": "synthetic: counter hash of (seed, global element index), uniform [-1, 1)"
                + (" (HARNESS TEST: parts share GPUs, not a reportable number)" if share else ""),
        "gflops": round(tps * cfg["flops"] / 1e9, 1),
        "config": {"workload": f"{cfg['name']}, batch=2^{int(np.log2(batch))} per GPU, out of place, device-resident, internal-layout spectrum",
                   "kernel": pa.kernel_name(setup), "batch_per_gpu": batch,
                   "sharding": "batch-split over the devices of ONE process: pffft_hip_transform_batch_multi, one shared setup, one stream per "
                               "device, no collective"},
        "single_process": True, "devices_seen": len(seen), "setup_devices": seen,
        "ms_per_step_fastest_device": round(min(per_dev_ms), 4), "ms_per_step_slowest_device": round(max(per_dev_ms), 4),
        "roofline": roofline(batch * cfg["bytes"], max(per_dev_ms) * 1e-3, None),
        "parity_vs_reference": parity, "source_hash": source_hash(),
    }
    print(json.dumps(out), flush=True)
    setup.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400, help="timed steps (default: >= 1 s of timed region on c2)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c2")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--batch-log2", type=int, default=None, help="override: transforms per GPU = 2^this (weak)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-extras", action="store_true", help="headline line only: skip the other configs / side rates")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help=argparse.SUPPRESS)
    ap.add_argument("--selftest-cpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--single-process", action="store_true",
                    help="one process drives all --gpus devices through pffft_hip_transform_batch_multi with one shared setup")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.single_process:
        return single_process_main(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if args.selftest_cpu:
        return fake_main(args, world, rank)

    import torch
    import pffft_amd as pa
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # PFFFT_BENCH_SHARE_GPU=1 (harness test only, never a reportable number): ranks share the visible GPUs round robin,
    # so that the multi-rank path can be exercised end to end on a 1-GPU box (with --backend gloo: RCCL refuses two ranks
    # on one device)
    share = os.environ.get("PFFFT_BENCH_SHARE_GPU") == "1"
    if torch.cuda.device_count() <= local_rank and not share:
        raise SystemExit(f"bench.py: rank {rank} has no GPU (local_rank {local_rank}, {torch.cuda.device_count()} visible)")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    from pffft_amd.sharding import combine_stats, shard_range
    timer = Timer(torch)
    cpu_s = 0.0 if args.no_cpu_baseline else args.cpu_seconds
    SEEDS = {"c2": 2, "c3": 3, "c5": 5}

    def run_fft(key, scaling, steps, warmup, batch_log2=None):
        cfg = CONFIGS[key]
        if scaling == "strong":
            total = 1 << cfg["total_log2"]
            first, batch = shard_range(total, rank, world)
            in_place = True
        else:
            batch = 1 << (batch_log2 if batch_log2 is not None else cfg["batch_log2"])
            first = rank * batch                # weak scaling: rank r holds global vectors [r * batch, (r + 1) * batch)
            in_place = False
        el, ks, kname, parity, first_ms = fft_config_run(torch, pa, dev, cfg, batch, first, steps, warmup, in_place, dist, rank,
                                                         SEEDS[key])
        st = combine_stats(el, float(batch), dist, dev)
        el, tot = st["elapsed_max"], st["units"]
        tps = tot * steps / el
        rec = {
            "value": round(tps / 1e6, 3), "unit": "M transforms/s", "ms_per_step": round(el / steps * 1e3, 4),
            "steps": steps, "warmup": warmup, "scaling": scaling, "dtype": cfg["dtype"], "gflops": round(tps * cfg["flops"] / 1e9, 1),
            "workload": f"{cfg['name']}, batch={'2^%d total / %d GPU(s), in place' % (cfg['total_log2'], world) if scaling == 'strong' else '2^%d per GPU, out of place' % int(np.log2(batch))}, "
                        "device-resident, internal-layout spectrum",
            "kernel": kname, "batch_per_gpu": batch, "first_global_vector_of_rank0": first,
            "roofline": roofline(batch * cfg["bytes"], ks, key + "_bytes_per_launch" if scaling == "weak" and batch == (1 << cfg["batch_log2"]) else None),
            "first_launches_ms": first_ms,
            "parity_vs_reference": parity,
        }
        if world > 1:
            rec["ranks_seen"] = st["ranks_seen"]
            rec["ms_per_step_fastest_rank"] = round(st["elapsed_min"] / steps * 1e3, 4)
            rec["ms_per_step_slowest_rank"] = round(st["elapsed_max"] / steps * 1e3, 4)
        return rec

    if args.config == "c4":
        if world > 1:
            raise SystemExit("bench.py: c4 (one 8 MiB signal) does not shard: replicas only (DESIGN.md §5)")
        res, add_cpu = fir_config(torch, pa, dev, timer, args.warmup)
        add_cpu(cpu_s)
        line = {"metric": "G output samples/s, pffastconv overlap-save FIR, 4096 taps", "value": res["value"],
                "unit": res["unit"], "n_gpus": 1, "steps": 40, "warmup": args.warmup,
                "ms_per_step": res["roofline"]["kernel_ms"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": res.pop("workload")}}
        line.update(res)
        print(json.dumps(line), flush=True)
        return

    head = run_fft(args.config, args.scaling, args.steps, args.warmup, args.batch_log2)
    cfg = CONFIGS[args.config]
    out = None
    if rank == 0:
        out = {
            "metric": "M transforms/s, batched N=%d %s-%s forward FFT (pffft_transform semantics)" % (
                cfg["N"], "complex" if cfg["tr"] else "real", "float" if cfg["dtype"] == "f32" else "double"),
            "value": head["value"], "unit": "M transforms/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": cfg["dtype"],
            "data": "synthetic: counter hash of (seed, global element index), uniform [-1, 1)"
                    + (" (HARNESS TEST: ranks share GPUs, not a reportable number)" if share else ""),
            "gflops": head["gflops"],
            "config": {"workload": head["workload"], "kernel": head["kernel"], "batch_per_gpu": head["batch_per_gpu"],
                       "sharding": "batch-split, no data-path collective; RCCL for the final MAX/SUM only"},
            "roofline": head["roofline"], "first_launches_ms": head["first_launches_ms"],
            "parity_vs_reference": head["parity_vs_reference"], "source_hash": source_hash(),
        }
        for k in ("ranks_seen", "ms_per_step_fastest_rank", "ms_per_step_slowest_rank"):
            if k in head:
                out[k] = head[k]
    extras, configs = {}, {}
    cpu_jobs = []        # CPU baselines run AFTER all GPU work: an idle GPU between configs drops its clocks
    if not args.no_extras and args.config == "c2" and args.scaling == "weak":
        if world == 1:
            # side rates of the headline kernel family + every other BASELINE config at its stated size
            setup = pa.Setup(1024, COMPLEX, np.float32)
            b = 1 << 20
            x = make_input(torch, dev, b, 2048, torch.float32, seed=2)
            y = torch.empty_like(x)
            for name, d, o in (("fwd_ordered", pa.FORWARD, True), ("inv_unordered", pa.BACKWARD, False),
                               ("inv_ordered", pa.BACKWARD, True)):
                ts = timer(lambda: setup.transform_batch(x, y, d, ordered=o), 20)
                extras[name + "_Mtps"] = round(b / ts / 1e6, 2)
            extras["torch_copy_GBps"] = round(b * 16384 / timer(lambda: y.copy_(x), 20) / 1e9, 1)
            ts = timer(lambda: setup.shift_transform_batch(x, 0.0137, 0.4, out=y, ordered=False), 20)
            extras["shift_fused_fwd_Mtps"] = round(b / ts / 1e6, 2)
            try:
                from pffft_amd import pfdsp
                xc, yc = x.view(torch.complex64).reshape(-1), y.view(torch.complex64).reshape(-1)
                extras["mixer_GBps"] = round(xc.numel() * 16 / timer(lambda: pfdsp.shift_device(xc, 0.0137, 0.4, out=yc), 20) / 1e9, 1)
            except Exception as e:  # the mixer library is its own .so; its absence must not hide the headline
                extras["mixer_error"] = str(e)[:200]
            setup.close()
            del x, y
            torch.cuda.empty_cache()
            # the other BASELINE configs: no warm-up beyond --warmup (their first launches are listed in first_launches_ms)
            for key, st in (("c3", 100), ("c5", 40)):
                try:
                    configs[key] = run_fft(key, "weak", st, args.warmup)

                    def job(key=key):
                        cb = cpu_baseline(CONFIGS[key], 10 if key == "c3" else 14, min(cpu_s, 5.0))
                        configs[key]["cpu_baseline"] = cb if cb else {"value": None, "unit": "M transforms/s", "cores": 0,
                                                                      "kind": "reference", "sample": "oracle/_ref not present"}
                    cpu_jobs.append(job)
                except Exception as e:
                    configs[key] = {"error": str(e)[:300]}
            try:
                configs["c5_strong_1gpu"] = run_fft("c5", "strong", 6, 2)
            except Exception as e:
                configs["c5_strong_1gpu"] = {"error": str(e)[:300]}
            try:
                configs["c4"], add_cpu = fir_config(torch, pa, dev, timer, args.warmup)
                cpu_jobs.append(lambda: add_cpu(min(cpu_s, 5.0)))
            except Exception as e:
                configs["c4"] = {"error": str(e)[:300]}
            try:
                configs["conv"] = conv_config(torch, pa, dev, timer, args.warmup)
            except Exception as e:
                configs["conv"] = {"error": str(e)[:300]}
            try:
                configs["batch_sweep"] = batch_sweep(torch, pa, dev, timer)
            except Exception as e:
                configs["batch_sweep"] = {"error": str(e)[:300]}
            try:
                Rr = None
                try:
                    from oracle import ref as oref
                    Rr = oref.get() if oref.available() else None
                except Exception:
                    Rr = None
                configs["sizes"] = sizes_table(torch, pa, dev, timer, Rr)
            except Exception as e:
                configs["sizes"] = {"error": str(e)[:300]}
        else:
            # BASELINE configs[4]: the double-precision config sharded over the same ranks, weak and strong
            for name, sc, st in (("c5_weak", "weak", 40), ("c5_strong", "strong", 8)):
                configs[name] = run_fft("c5", sc, st, args.warmup if sc == "weak" else 2)   # collective inside: every rank runs it
    if rank == 0:
        out.update(extras)
        if world == 1 and cpu_s > 0:
            for job in cpu_jobs:
                try:
                    job()
                except Exception as e:
                    out.setdefault("cpu_baseline_errors", []).append(str(e)[:200])
            cb = cpu_baseline(cfg, 15 if cfg["N"] <= 1024 else 10, cpu_s)
            out["cpu_baseline"] = cb if cb else {"value": None, "unit": "M transforms/s", "cores": 0, "kind": "reference",
                                                 "sample": "oracle/_ref not present on this box"}
        if configs:
            # The side configs in full (first launches, CPU baselines, the whole sizes table) go to STDERR as one JSON line; the
            # one line on stdout carries a compact summary of each INSIDE `roofline` - the dict the driver's record keeps whole
            # (r03: `configs` was dropped from the stored record and the stdout tail began inside C5).
            print(json.dumps({"detail": configs}), file=sys.stderr, flush=True)
            summ = {}
            for key in ("c3", "c5", "c5_strong_1gpu", "c5_weak", "c5_strong"):
                c = configs.get(key)
                if isinstance(c, dict) and "roofline" in c:
                    par = c.get("parity_vs_reference")
                    summ[key] = {"frac": c["roofline"]["frac"], "kernel_ms": c["roofline"]["kernel_ms"], "value_Mtps": c["value"],
                                 "first_launch_ms": (c.get("first_launches_ms") or [None])[0],
                                 "parity_max_rel_err": par.get("max_rel_err") if isinstance(par, dict) else par,
                                 "cpu_Mtps": (c.get("cpu_baseline") or {}).get("value")}
                elif isinstance(c, dict):
                    summ[key] = c
            c4 = configs.get("c4")
            if isinstance(c4, dict) and "roofline" in c4:
                summ["c4"] = {"long_frac": c4["roofline"]["frac"], "long_frac_events": c4["roofline"].get("frac_events", c4["roofline"]["frac"]),
                              "long_kernel_ms": c4["roofline"]["kernel_ms"], "long_kernel_ms_trace": c4["roofline"].get("kernel_ms_trace"),
                              "batch_frac": c4.get("batch_frac"), "single_call_us": c4.get("single_call_us"),
                              "single_call_frac": c4.get("single_call_frac"),
                              "bound": c4["roofline"].get("bound"), "lds_frac": (c4.get("roofline_lds") or {}).get("frac"),
                              "valu_frac_measured_peak": c4["roofline_valu"]["frac"], "valu_frac_spec_peak": c4["roofline_valu"]["frac_spec"],
                              "parity_err_over_range": c4.get("parity_max_err_over_range"),
                              "cpu_Gsps": (c4.get("cpu_baseline") or {}).get("value")}
            elif isinstance(c4, dict):
                summ["c4"] = c4
            cv = configs.get("conv")
            if isinstance(cv, dict) and "roofline" in cv:
                summ["conv"] = {"frac": cv["roofline"]["frac"], "kernel_ms": cv["roofline"]["kernel_ms"], "value_Mps": cv["value"],
                                "composed_Mps": cv.get("composed_three_launches_Mps"), "parity_max_rel_err": cv.get("parity_max_rel_err")}
            elif isinstance(cv, dict):
                summ["conv"] = cv
            bs = configs.get("batch_sweep")
            if isinstance(bs, dict):
                summ["batch_sweep"] = {k: v for k, v in bs.items() if k in ("c2", "c5", "error", "protocol")}
            sz = configs.get("sizes")
            if isinstance(sz, dict) and "f32_complex" in sz:
                summ["sizes"] = {"[min, mean, share >= 0.70] lds-resident / [min, mean] beyond (best of two runs per entry)": sizes_summary(sz),
                                 "mean_of_two_runs": sz.get("mean_of_two_runs"),
                                 "parity_worst_rel_err": sz.get("parity_worst_rel_err")}
            elif isinstance(sz, dict):
                summ["sizes"] = sz
            out["roofline"]["configs_summary"] = summ
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
