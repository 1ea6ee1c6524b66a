"""Model the LDS accesses of fft_tiled.h exchanges and search PAD0 / PADN per configuration."""
import sys, itertools
import numpy as np
sys.path.insert(0, 'tools')
import lds_sim

def ns(rads, s):
    p = 1
    for i in range(s): p *= rads[i]
    return p

def jmap(t, u, TPT, B, VEC):
    if VEC == 2 and B % 2 == 0:
        return 2*t + (u & 1) + 2*TPT*(u >> 1)
    return t + TPT*u

def evaluate(n, TPT, rads, esz, PAD0, PADN, verbose=False):
    E = n // TPT
    VEC = 16 // esz
    R0 = rads[0]
    def phys_nat(P): return P + PADN*(P >> 6)
    def phys_trn(P): return (P & (R0-1))*(n//R0 + PAD0) + (P // R0)
    wk, rk = ("w64", "r64") if esz == 8 else ("w128", "r128")
    tot = 0; ideal = 0; detail = []
    lanes = np.arange(64)
    # a wave = 64 consecutive threads of the workgroup: thread index -> (slot, t)
    for S in range(len(rads)-1):
        R, Ns, B = rads[S], ns(rads, S), E // rads[S]
        R2, B2 = rads[S+1], E // rads[S+1]
        trn = (S == 0)
        ph = phys_trn if trn else phys_nat
        w = r = wi = ri = 0
        for wave in range(max(1, TPT // 64)):
            tid = lanes + 64*wave
            t = tid % TPT; slot = tid // TPT
            img_off = slot * (max(n + PADN*(n//64), R0*(n//R0+PAD0)) + 8)
            for u in range(B):
                j = jmap(t, u, TPT, B, VEC)
                base = (j // Ns)*(Ns*R) + (j & (Ns-1))
                for d in range(R):
                    P = base + d*Ns
                    addr = (np.array([ph(int(p)) for p in P]) + img_off) * esz
                    w += lds_sim.cycles(wk, addr); wi += lds_sim.ideal(wk)
            for u in range(B2):
                j = jmap(t, u, TPT, B2, VEC)
                for q in range(R2):
                    P = j + q*(n//R2)
                    addr = (np.array([ph(int(p)) for p in P]) + img_off) * esz
                    r += lds_sim.cycles(rk, addr); ri += lds_sim.ideal(rk)
        detail.append((S, w, wi, r, ri))
        tot += w + r; ideal += wi + ri
    return tot, ideal, detail

CONFIGS = {
    512: (32, [8, 8, 8]), 1024: (64, [8, 16, 8]), 2048: (128, [8, 4, 8, 8]), 4096: (128, [16, 16, 16]),
    8192: (512, [8, 8, 16, 8]), 16384: (512, [16, 8, 8, 16]),
}
if __name__ == "__main__":
    which = [int(a) for a in sys.argv[1:]] or sorted(CONFIGS)
    for n in which:
        TPT, rads = CONFIGS[n]
        for esz in (8, 16):
            base = evaluate(n, TPT, rads, esz, 0, 0)
            best = None
            for PAD0 in (0, 1, 2, 4, 8, 16):
                for PADN in (0, 1, 2, 4, 8):
                    if esz == 8 and (PAD0 % 1 or PADN % 1): continue
                    tot, ideal, det = evaluate(n, TPT, rads, esz, PAD0, PADN)
                    if best is None or tot < best[0]:
                        best = (tot, ideal, PAD0, PADN, det)
            print(f"n={n} esz={esz}: unpadded {base[0]}/{base[1]}  best {best[0]}/{best[1]} PAD0={best[2]} PADN={best[3]}  detail(S,w,wi,r,ri)={best[4]}")
