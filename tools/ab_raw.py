"""A/B of two builds of libpffft_hip.so through the reference-era symbols only (pffft[d]_new_setup, pffft[d]_hip_transform_batch), so that an OLD build
that lacks later entries can be one side:   python tools/ab_raw.py libA.so libB.so 4000:r:f32 9216:c:f32 ...   (fractions of 8 TB/s, 1 GiB per
launch, four direction x layout combinations, min of 3 x 10 launches, A and B alternating per size in one process)."""
import ctypes as C, sys
import numpy as np, torch

def load(path):
    L = C.CDLL(path)
    for p in ("pffft", "pffftd"):
        getattr(L, p + "_new_setup").restype = C.c_void_p; getattr(L, p + "_new_setup").argtypes = [C.c_int, C.c_int]
        getattr(L, p + "_destroy_setup").argtypes = [C.c_void_p]
        f = getattr(L, p + "_hip_transform_batch"); f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    return L

def frac(L, N, real, dbl):
    p = "pffftd" if dbl else "pffft"
    s = getattr(L, p + "_new_setup")(N, 0 if real else 1)
    assert s, (N, real, dbl)
    scal = N if real else 2 * N
    isz = 8 if dbl else 4
    batch = max(1, (1 << 30) // (scal * isz))
    x = torch.rand((batch, scal), device="cuda", dtype=torch.float64 if dbl else torch.float32) * 2 - 1
    y = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    f = getattr(L, p + "_hip_transform_batch")
    out = []
    for d in (0, 1):
        for o in (1, 0):
            run = lambda: f(s, x.data_ptr(), y.data_ptr(), batch, d, o, st)
            for _ in range(5): assert run() == 0
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10): run()
                b.record(); torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b) / 10 * 1e-3)
            out.append(2 * x.numel() * isz / best / 8e12)
    getattr(L, p + "_destroy_setup")(s)
    return out

A, B = load(sys.argv[1]), load(sys.argv[2])
for spec in sys.argv[3:]:
    N, tr, dt = spec.split(":")
    for rep in range(2):
        for tag, L in (("A", A), ("B", B)):
            r = frac(L, int(N), tr == "r", dt == "f64")
            print(f"{spec:>14} {tag}{rep}: " + "  ".join(f"{v:.3f}" for v in r) + f"   mean {sum(r) / 4:.3f}", flush=True)
