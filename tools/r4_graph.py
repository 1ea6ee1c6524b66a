"""The stated C4 call (2^20 samples, 4096 taps) and N = 1024 transforms of small batches replayed from a captured HIP graph (development tool):
the library's launches are capturable once the lazily built tables exist (no allocation, no synchronisation on the launch path)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pffft_amd as pa


def per_call(f, n):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3          # us


def main():
    rng = np.random.default_rng(4)
    for taps, L in ((4096, 1 << 20), (2048, 1 << 19), (1500, 300001)):
        h = rng.uniform(-1, 1, taps).astype(np.float32)
        fc = pa.FastConv(h, 0, 0)
        x = torch.rand(L, device="cuda") * 2 - 1
        y = torch.empty_like(x)
        ref, n = fc.apply(x, True, out=y); ref = ref.clone()
        plain = min(per_call(lambda: fc.apply(x, True, out=y), 300) for _ in range(3))
        res = [f"plain {plain:.2f} us"]
        for k in (1, 16):
            s = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s):
                fc.apply(x, True, out=y)                              # warm (tables) on the capture stream
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=s):
                    for _ in range(k): fc.apply(x, True, out=y)
            y.zero_()
            g.replay(); torch.cuda.synchronize()
            ok = torch.equal(y[:n], ref[:n])
            t = min(per_call(g.replay, 300 // k + 20) for _ in range(3)) / k
            res.append(f"graph of {k}: {t:.2f} us per call ({'same values' if ok else 'VALUES DIFFER'})")
        print(f"FIR {taps} taps on {L} samples: " + ", ".join(res), flush=True)
        fc.close()

    s1 = pa.Setup(1024, pa.COMPLEX)
    for B in (256, 4096):
        x = torch.rand(B, 2048, device="cuda"); y = torch.empty_like(x)
        s1.transform_batch(x, y, pa.FORWARD, False)
        plain = min(per_call(lambda: s1.transform_batch(x, y, pa.FORWARD, False), 300) for _ in range(3))
        st = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            s1.transform_batch(x, y, pa.FORWARD, False); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=st):
                for _ in range(16): s1.transform_batch(x, y, pa.FORWARD, False)
        t = min(per_call(g.replay, 40) for _ in range(3)) / 16
        print(f"N=1024 complex, batch {B}: plain {plain:.2f} us, graph of 16: {t:.2f} us per call", flush=True)


if __name__ == "__main__":
    main()
