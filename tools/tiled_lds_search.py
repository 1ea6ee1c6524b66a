"""Model the LDS accesses of fft_tiled.h exchanges (incl. the 16-byte pair accesses) and search PAD0 / PADN."""
import sys
import numpy as np
sys.path.insert(0, 'tools')
import lds_sim

def ns(rads, s):
    p = 1
    for i in range(s): p *= rads[i]
    return p

def evaluate(n, TPT, rads, esz, PAD0, PADN, real_dir=None):
    """real_dir: None complex, 'f' real forward (last stage symmetric), 'b' real backward (stage 0 symmetric)"""
    E = n // TPT; VEC = 16 // esz; R0 = rads[0]; NS = len(rads)
    sym = {None: -1, 'f': NS - 1, 'b': 0}[real_dir]
    def phys_nat(P): return P + PADN * (P >> 6)
    IMG = max(n + PADN * (n // 64), R0 * (n // R0 + PAD0)) + 8
    def pair(S): return VEC == 2 and (E // rads[S]) % 2 == 0 and S != sym and PAD0 % 2 == 0 and PADN % 2 == 0
    def jm(S, t, u):
        B = E // rads[S]
        if S == sym:
            nb = n // rads[S]
            return np.where(u == 0, t, np.where(t == 0, nb // 2, nb - t)) if np.isscalar(u) else None
        if VEC == 2 and B % 2 == 0: return 2 * t + (u & 1) + 2 * TPT * (u >> 1)
        return t + TPT * u
    tot = ideal = 0; detail = []
    lanes = np.arange(64)
    for S in range(NS - 1):
        R, Ns_, B = rads[S], ns(rads, S), E // rads[S]
        R2, B2 = rads[S + 1], E // rads[S + 1]
        w = wi = r = ri = 0
        for wave in range(max(1, TPT // 64)):
            tid = lanes + 64 * wave; t = tid % TPT; off = (tid // TPT) * IMG
            # write
            pw = pair(S) and (S == 0 or Ns_ >= 2)
            for u in (range(0, B, 2) if pw else range(B)):
                j = jm(S, t, u)
                for d in range(R):
                    if S == 0: a = j + d * (n // R0 + PAD0)
                    else:
                        Ha = (j // Ns_) * (Ns_ * R) + (j % Ns_)
                        a = Ha + PADN * (Ha >> 6) + d * Ns_ + PADN * ((d * Ns_) >> 6)
                    kind = ("w128" if pw else "w64") if esz == 8 else "w128"
                    w += lds_sim.cycles(kind, (a + off) * esz); wi += lds_sim.ideal(kind)
            pr = S > 0 and pair(S + 1)
            for u in (range(0, B2, 2) if pr else range(B2)):
                j = jm(S + 1, t, u)
                for q in range(R2):
                    if S == 0: a = (j % R0) * (n // R0 + PAD0) + j // R0 + q * (n // (R2 * R0))
                    else: a = j + PADN * (j >> 6) + q * (n // R2) + PADN * ((q * (n // R2)) >> 6)
                    kind = ("r128" if pr else "r64") if esz == 8 else "r128"
                    r += lds_sim.cycles(kind, (a + off) * esz); ri += lds_sim.ideal(kind)
        detail.append((S, w, wi, r, ri)); tot += w + r; ideal += wi + ri
    return tot, ideal, detail

CONFIGS = {512: (32, [8, 8, 8]), 1024: (64, [8, 16, 8]), 2048: (128, [8, 4, 8, 8]), 4096: (256, [8, 8, 8, 8]),
           8192: (512, [8, 8, 16, 8]), 16384: (1024, [8, 16, 16, 8])}
if __name__ == "__main__":
    which = [int(a) for a in sys.argv[1:]] or sorted(CONFIGS)
    for n in which:
        TPT, rads = CONFIGS[n]
        for esz in (8, 16):
            for rd in (None, 'f'):
                best = None
                for PAD0 in (0, 2, 4, 6, 8, 12, 16):
                    for PADN in (0, 2, 4, 6, 8, 10, 12, 16):
                        tot, ideal, det = evaluate(n, TPT, rads, esz, PAD0, PADN, rd)
                        if best is None or tot < best[0]: best = (tot, ideal, PAD0, PADN, det)
                cur = evaluate(n, TPT, rads, esz, 4, {512: 4, 1024: 4}.get(n, 1) if esz == 8 else 0, rd)
                print(f"n={n} esz={esz} real={rd}: current {cur[0]}/{cur[1]}  best {best[0]}/{best[1]} PAD0={best[2]} PADN={best[3]} {best[4]}")
