"""Generate tests/golden/pfdsp_golden.npz from the REAL reference mixers (oracle/_ref/libpfdsp_ref.so, built from
/root/reference/src/pf_mixer.cpp by oracle/Makefile).  Run in the dev container:

    make -C oracle && python tests/golden/make_pfdsp_golden.py

The reference has no stored vectors for its mixers (benchmarks/bench_mixers.cpp only times them); these fixtures are the
outputs, returned phases and advanced state structs of the reference's own object code on one seeded 512-sample block per
algorithm and rate, two chained calls of 256 — committed so that the numpy restatement (oracle/pfdsp_oracle.py) and the
HIP path can be checked where /root/reference and oracle/_ref are absent.  ~100 KiB.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pfdsp_ref  # noqa: E402
from pffft_amd import pfdsp  # noqa: E402

RATES = (0.0137, -0.21)
PH0 = 0.4
N = 512


def state_of(m):
    a = m.algo
    if a in ("math", "table", "addfast", "unroll"):
        return np.array([m.phase], np.float64)
    if a == "limited_unroll":
        return np.array([m.data.complex_phase.i, m.data.complex_phase.q], np.float64)
    if a.startswith("limited_unroll_"):
        return np.array(list(m.data.phase_state_i[:]) + list(m.data.phase_state_q[:]), np.float64)
    k = 8 if a == "recursive_osc" else 4
    return np.array(list(m.data.u_cos[:k]) + list(m.data.v_sin[:k]) + [m.conf.k1, m.conf.k2], np.float64)


def main():
    R = pfdsp_ref.get()
    rng = np.random.default_rng(20260926)
    x = (rng.uniform(-1, 1, N) + 1j * rng.uniform(-1, 1, N)).astype(np.complex64)
    out = {"x": x, "rates": np.array(RATES), "ph0": np.array([PH0])}
    for ri, rate in enumerate(RATES):
        for algo in pfdsp.ALGOS:
            if algo == "table":
                continue   # reference bug: quadrant-resolution oscillator (src/pf_mixer.cpp:202), not a fixture worth pinning
            m = pfdsp.Mixer(algo, rate, PH0, abi=R)
            y0 = m(np.ascontiguousarray(x[:N // 2]))
            s0 = state_of(m)
            y1 = m(np.ascontiguousarray(x[N // 2:]))
            out[f"{algo}_r{ri}_y"] = np.concatenate([y0, y1])
            out[f"{algo}_r{ri}_state_mid"] = s0
            out[f"{algo}_r{ri}_state_end"] = state_of(m)
            m.close()
    path = os.path.join(ROOT, "tests", "golden", "pfdsp_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
