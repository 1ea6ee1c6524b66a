"""Generate tests/golden/pffft_golden.npz from the REAL reference (oracle/_ref/libpffft_ref.so, built
from /root/reference by oracle/Makefile).  Run in the dev container:

    make -C oracle && python tests/golden/make_golden.py

The reference holds no stored golden vectors; these fixtures are outputs of the
reference's own code on seeded inputs, committed so that the oracle restatement and the HIP path can
be checked where /root/reference and oracle/_ref are absent.  Kept small (a few hundred KiB).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref as oref  # noqa: E402

CASES = [  # (dtype, transform, N) — covers radix 2/3/4/5 mixes (benchmarks/bench_pffft.c:445) + BASELINE configs
    ("f32", oref.REAL, 64), ("f32", oref.REAL, 96), ("f32", oref.REAL, 160), ("f32", oref.REAL, 480),
    ("f32", oref.REAL, 1024), ("f32", oref.REAL, 4000), ("f32", oref.REAL, 8192), ("f32", oref.REAL, 16384),
    ("f32", oref.COMPLEX, 16), ("f32", oref.COMPLEX, 48), ("f32", oref.COMPLEX, 80), ("f32", oref.COMPLEX, 128),
    ("f32", oref.COMPLEX, 1024), ("f32", oref.COMPLEX, 2592),
    ("f64", oref.REAL, 64), ("f64", oref.REAL, 1024), ("f64", oref.COMPLEX, 64), ("f64", oref.COMPLEX, 1024),
    ("f64", oref.COMPLEX, 96),
]


def gen_input(seed, n, dtype):
    return np.random.default_rng(seed).uniform(-1.0, 1.0, n).astype(dtype)


def main():
    R = oref.get()
    out = {}
    for i, (dt, tr, N) in enumerate(CASES):
        dtype = np.float32 if dt == "f32" else np.float64
        s = R.setup(N, tr, dtype)
        nf = s.nfloats
        key = f"{dt}_{'r' if tr == oref.REAL else 'c'}{N}"
        x = gen_input(1000 + i, nf, dtype)
        fo = s.transform_ordered(x, oref.FORWARD)
        fu = s.transform_unordered(x, oref.FORWARD)
        out[key + "_x"] = x
        out[key + "_fwd_ordered"] = fo
        out[key + "_fwd_unordered"] = fu
        out[key + "_bwd_ordered"] = s.transform_ordered(fo, oref.BACKWARD)
        out[key + "_bwd_unordered"] = s.transform_unordered(fu, oref.BACKWARD)
        if N <= 1024:
            out[key + "_perm"] = s.zreorder(np.arange(nf, dtype=dtype), oref.FORWARD).astype(np.int32)
            b = s.transform_unordered(gen_input(2000 + i, nf, dtype), oref.FORWARD)
            acc = gen_input(3000 + i, nf, dtype)
            out[key + "_zc_b"] = b
            out[key + "_zc_acc0"] = acc
            out[key + "_zc_accumulate"] = s.zconvolve(fu, b, acc, 0.25, True)
            out[key + "_zc_no_accu"] = s.zconvolve(fu, b, acc, 0.25, False)
        s.close()
    # fast convolution (src/pffastconv.c) — reference pattern of tests/test_pffastconv.c:539,565-571 and a random case
    fc = [("ramp", 3000, 129, 0, 0), ("rand", 5000, 64, 512, 0), ("cplx2", 1500, 33, 0, 1), ("cplx1", 1500, 33, 0, 17),
          ("corr", 2000, 48, 0, 64)]
    for name, L, taps, blk, flags in fc:
        cpl = 2 if flags & 1 else 1
        if name == "ramp":
            x = (np.arange(L * cpl) % 4093).astype(np.float32)
            h = np.array([(-1.0, 1.0, 0.5)[j % 3] for j in range(taps)], dtype=np.float32)
        else:
            x = gen_input(4000 + L, L * cpl, np.float32)
            h = gen_input(5000 + taps, taps, np.float32)
        for flush in (0, 1):
            y, n, bl = R.fastconv(x, h, blk, flags, flush)
            k = f"fc_{name}_flush{flush}"
            out[k + "_x"], out[k + "_h"], out[k + "_y"] = x, h, y
            out[k + "_meta"] = np.array([L, taps, blk, flags, flush, n, bl], dtype=np.int64)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pffft_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()
