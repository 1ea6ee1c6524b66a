#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X pffft drop-in.

Metric (BASELINE.json): M transforms/s (+ GFLOPS + % of HBM roofline) on batched N=1024
complex-float forward FFTs — configs[1] ("N=1024 complex float fwd+inv, batch=1M, 1xMI355X").
A "step" is one pass of the hot path (pffft_transform semantics, internal-layout spectrum) over the
whole device-resident batch = ONE kernel launch through the C ABI (pffft_hip_transform_batch).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: the batch shards across ranks with no data-path collective; every rank
transforms its own 2^20 vectors (weak scaling); RCCL is used only to combine the final number
(all_reduce MAX of elapsed, SUM of transforms), bracketed by barriers.

Prints ONE JSON line on rank 0.  `value` = whole-job M transforms/s with inputs resident in HBM.
`roofline` prices the kernel's ALGORITHMIC bytes (16 KiB per transform: 8 KiB read + 8 KiB written,
SURVEY.md §8d) against 8 TB/s.  `cpu_baseline` times the real reference (oracle/_ref) on the host
cores of this box on a bounded sample of the same workload (test infrastructure, never the product).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FFT = 1024
BYTES_PER_TRANSFORM = 2 * N_FFT * 2 * 4      # 8 KiB in + 8 KiB out (algorithmic, twiddles not counted)
FLOPS_PER_TRANSFORM = 5 * N_FFT * 10         # 5 N log2 N (benchmarks/bench_pffft.c:606 convention)
HBM_PEAK = 8.0e12                            # MI355X HBM3E spec (MI355X_MICROARCH.md)


def cpu_baseline(sample_log2: int, target_s: float):
    """Time oracle/_ref (the reference's own SIMD path) on this host: loop of pffft_transform calls."""
    from oracle import ref as oref
    so = os.path.join(ROOT, "oracle", "_ref", "libcpubase.so")
    if not (oref.available() and os.path.exists(so)):
        return None
    lib = C.CDLL(so)
    lib.cpu_baseline_run.restype = C.c_double
    lib.cpu_baseline_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_long, C.c_int, C.c_int]
    R = oref.get()
    batch = 1 << sample_log2
    a = R.f32
    x = a.empty(batch * 2 * N_FFT)
    x[:] = np.random.default_rng(2).uniform(-1, 1, x.size).astype(np.float32)
    y = a.empty(batch * 2 * N_FFT)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass

    def run(threads, reps):
        return lib.cpu_baseline_run(N_FFT, 1, 0, 0, 0, x.ctypes.data, y.ctypes.data, batch, reps, threads)

    run(min(cores, 8), 1)  # warm-up (page faults, caches)
    t1 = run(1, 2)
    one_thread = 2 * batch / t1
    # the affinity mask can overstate what a container may use: pick the thread count that is
    # actually fastest on a short calibration, then spend the time budget there
    cands = sorted({c for c in (8, 16, 32, 64, 96, 128, 192, 256, cores) if c <= cores} | {min(cores, 4)})
    best_t, best_rate = 1, one_thread
    for th in cands:
        r = 4
        rate = r * batch / run(th, r)
        if rate > best_rate:
            best_t, best_rate = th, rate
    cores = best_t
    # spend ~target_s in chunks (the sustained rate under full load is lower than a short calibration suggests)
    chunk = max(1, int(0.5 * best_rate / batch))
    reps, tall = 0, 0.0
    while tall < target_s and reps < 1000000:
        tall += run(cores, chunk)
        reps += chunk
    all_cores = batch * reps / tall
    return {
        "value": round(all_cores / 1e6, 4), "unit": "M transforms/s", "cores": cores, "kind": "reference",
        "sample": f"{reps} x 2^{sample_log2} transforms (N=1024 cplx f32 fwd, pffft_transform loop, "
                  f"{tall:.1f} s wall, {cores} threads sharing one setup)",
        "one_thread_value": round(one_thread / 1e6, 4), "simd_arch": a.simd_arch().decode(),
        "gflops": round(all_cores * FLOPS_PER_TRANSFORM / 1e9, 2),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-log2", type=int, default=20, help="transforms per GPU = 2^this (default 1M)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-extras", action="store_true", help="skip the inverse / ordered side measurements")
    args = ap.parse_args()

    import torch
    import pffft_amd as pa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    batch = 1 << args.batch_log2
    setup = pa.Setup(N_FFT, pa.COMPLEX, np.float32)
    gen = torch.Generator(device=dev)
    gen.manual_seed(2 + rank)
    x = torch.rand(batch, 2 * N_FFT, device=dev, generator=gen) * 2 - 1   # uniform [-1, 1), generated on device
    y = torch.empty_like(x)

    def step():
        setup.transform_batch(x, y, pa.FORWARD, ordered=False)

    # ---- parity spot check (outside the timed region) against the real reference when it travelled ----
    parity = None
    step()
    torch.cuda.synchronize()
    try:
        from oracle import ref as oref
        if oref.available():
            rs = oref.get().setup(N_FFT, oref.COMPLEX, np.float32)
            idx = [0, 1, batch // 2, batch - 1]
            want = rs.batch(x[idx].cpu().numpy(), oref.FORWARD, False)
            got = y[idx].cpu().numpy()
            parity = float(np.abs(got - want).max() / np.abs(want).max())
    except Exception as e:  # the checker is optional here; tests/ are the parity gate
        parity = f"unchecked: {e}"

    # W warm-up steps, then exactly K timed steps bracketed by barrier + synchronize on both sides
    # (pffft_amd/sharding.py); HIP events on the launch stream give the kernel's own average duration.
    from pffft_amd.sharding import combine, timed_steps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    count = [0]

    def timed_step():
        if count[0] == args.warmup:
            e0.record()
        step()
        count[0] += 1
        if count[0] == args.warmup + args.steps:
            e1.record()

    elapsed = timed_steps(timed_step, args.steps, args.warmup, dist, torch.cuda.synchronize)
    kernel_s = e0.elapsed_time(e1) * 1e-3 / args.steps
    elapsed, total = combine(elapsed, float(batch), dist, dev)

    extras = {}
    if not args.no_extras and rank == 0 and world == 1:   # side measurements only in the single-GPU run: no rank waits on another
        def t_of(fn, reps=5):
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) * 1e-3 / reps
        for name, d, o in (("fwd_ordered", pa.FORWARD, True), ("inv_unordered", pa.BACKWARD, False),
                           ("inv_ordered", pa.BACKWARD, True)):
            ts = t_of(lambda: setup.transform_batch(x, y, d, ordered=o))
            extras[name + "_Mtps"] = round(batch / ts / 1e6, 2)
        ts = t_of(lambda: y.copy_(x))
        extras["torch_copy_GBps"] = round(batch * BYTES_PER_TRANSFORM / ts / 1e9, 1)
        # the other BASELINE.json configs (parity-test cases, reported here as side measurements only): fraction of the
        # 8 TB/s HBM roofline on algorithmic bytes (2 x vector bytes per transform), and the C4 FIR call time
        try:
            del_later = []
            def side(N, tr, dtype, b):
                st = pa.Setup(N, tr, dtype)
                tdt = torch.float32 if dtype == np.float32 else torch.float64
                xi = torch.rand(b, st.vec_scalars, device=dev, dtype=tdt) * 2 - 1
                yo = torch.empty_like(xi)
                tt = t_of(lambda: st.transform_batch(xi, yo, pa.FORWARD, ordered=False))
                fr = 2 * xi.numel() * xi.element_size() / tt / HBM_PEAK
                st.close(); del xi, yo
                return round(fr, 4), round(b / tt / 1e6, 3)
            extras["c3_real16384_f32_fwd_frac"], extras["c3_Mtps"] = side(16384, pa.REAL, np.float32, 1 << 14)
            extras["c5_cplx1024_f64_fwd_frac"], extras["c5_Mtps"] = side(1024, pa.COMPLEX, np.float64, 1 << 18)
            sig = torch.rand(1 << 20, device=dev) * 2 - 1
            taps = np.random.default_rng(4).uniform(-1, 1, 4096).astype(np.float32)
            fc = pa.FastConv(taps, 0, 0)
            yo = torch.empty_like(sig)
            extras["c4_fir_us_per_call"] = round(t_of(lambda: fc.apply(sig, True, out=yo), reps=50) * 1e6, 2)
            fc.close(); del sig, yo
            torch.cuda.empty_cache()
        except Exception as e:
            extras["side_configs_error"] = str(e)[:200]
        # the PFDSP mixer fused into the load stage of the same transform, and the mixer alone
        ts = t_of(lambda: setup.shift_transform_batch(x, 0.0137, 0.4, out=y, ordered=False))
        extras["shift_fused_fwd_Mtps"] = round(batch / ts / 1e6, 2)
        try:
            from pffft_amd import pfdsp
            xc, yc = x.view(torch.complex64).reshape(-1), y.view(torch.complex64).reshape(-1)
            ts = t_of(lambda: pfdsp.shift_device(xc, 0.0137, 0.4, out=yc))
            extras["mixer_GBps"] = round(xc.numel() * 16 / ts / 1e9, 1)
        except Exception as e:  # the mixer library is its own .so; its absence must not hide the headline
            extras["mixer_GBps"] = None
            extras["mixer_error"] = str(e)[:200]

    if rank == 0:
        tps = total * args.steps / elapsed
        achieved = batch * BYTES_PER_TRANSFORM / kernel_s
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written by tools/pmc_summary.py from rocprofv3 --pmc
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("c1024_fwd_unordered_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "M transforms/s, batched N=1024 complex-float forward FFT (pffft_transform semantics)",
            "value": round(tps / 1e6, 3), "unit": "M transforms/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "gflops": round(tps * FLOPS_PER_TRANSFORM / 1e9, 1),
            "config": {"workload": "BASELINE configs[1]: N=1024 complex float forward, batch=2^%d per GPU, "
                                   "device-resident, out-of-place, internal-layout spectrum" % args.batch_log2,
                       "kernel": pa.kernel_name(setup), "batch_per_gpu": batch, "sharding": "batch-split, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK, 4), "traffic": traffic,
                         "kernel_ms": round(kernel_s * 1e3, 4), "algorithmic_bytes_per_launch": batch * BYTES_PER_TRANSFORM},
            "parity_max_rel_err_vs_reference": parity,
        }
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(15, args.cpu_seconds)
            out["cpu_baseline"] = cb if cb else {"value": None, "unit": "M transforms/s", "cores": 0, "kind": "reference",
                                                 "sample": "oracle/_ref not present on this box"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
