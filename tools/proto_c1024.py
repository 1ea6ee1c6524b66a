"""numpy prototype of the wave-per-transform N=1024 complex kernel (8 x 16 x 8, DIF).
Checks the index algebra against numpy.fft and evaluates LDS bank conflicts of each
exchange with tools/lds_sim.py."""
import numpy as np, sys, itertools
sys.path.insert(0, 'tools')
import lds_sim

N = 1024
W = np.exp(-2j*np.pi*np.arange(N)/N)
def dft(a, axis):  # forward DFT along axis
    R = a.shape[axis]
    M = np.exp(-2j*np.pi*np.outer(np.arange(R), np.arange(R))/R)
    return np.moveaxis(np.tensordot(M, np.moveaxis(a, axis, 0), axes=(1, 0)), 0, axis)

def run(x, S1=136, S2=136, conflicts=None):
    L = np.arange(64)
    # load: a[j][e][L] = x[128 j + 2L + e]
    a = np.empty((8, 2, 64), complex)
    for j in range(8):
        for e in range(2):
            a[j, e] = x[128*j + 2*L + e]
    y = dft(a, 0)                      # y[k1][e][L]
    for k1 in range(8):
        for e in range(2):
            y[k1, e] *= W[(k1*(2*L+e)) % N]
    lds = np.zeros(8*max(S1,S2)+64, complex)
    # X1 write (b128: 2 complex)
    for k1 in range(8):
        addr = k1*S1 + 2*L
        if conflicts is not None: conflicts.append(("X1w", lds_sim.cycles("w128", addr*8), lds_sim.ideal("w128")))
        lds[addr] = y[k1, 0]; lds[addr+1] = y[k1, 1]
    # X1 read (b64): lane (k1'=L>>3, c=L&7), a=0..15
    k1p, c = L >> 3, L & 7
    a2 = np.empty((16, 64), complex)
    for aa in range(16):
        addr = k1p*S1 + 8*aa + c
        if conflicts is not None: conflicts.append(("X1r", lds_sim.cycles("r64", addr*8), lds_sim.ideal("r64")))
        a2[aa] = lds[addr]
    z = dft(a2, 0)                     # z[ka][L]
    for ka in range(16):
        z[ka] *= W[(8*c*ka) % N]
    # X2 write b64: A2(k1,ka,c) = k1*S2 + ka*8 + c
    lds2 = np.zeros(8*S2+64, complex)
    for ka in range(16):
        addr = k1p*S2 + ka*8 + c
        if conflicts is not None: conflicts.append(("X2w", lds_sim.cycles("w64", addr*8), lds_sim.ideal("w64")))
        lds2[addr] = z[ka]
    # X2 read b128: lane L, e: k1 = 2(L&3)+e, ka = L>>2, c0 = 0,2,4,6
    a3 = np.empty((8, 2, 64), complex)
    for e in range(2):
        k1 = 2*(L & 3) + e; ka = L >> 2
        for c0 in range(0, 8, 2):
            addr = k1*S2 + ka*8 + c0
            if conflicts is not None: conflicts.append(("X2r", lds_sim.cycles("r128", addr*8), lds_sim.ideal("r128")))
            a3[c0, e] = lds2[addr]; a3[c0+1, e] = lds2[addr+1]
    Xr = dft(a3, 0)                    # Xr[kc][e][L] = X[2L+e+128kc]
    X = np.empty(N, complex)
    for kc in range(8):
        for e in range(2):
            X[2*L + e + 128*kc] = Xr[kc, e]
    return X

rng = np.random.default_rng(1)
x = rng.standard_normal(N) + 1j*rng.standard_normal(N)
conf = []
X = run(x, conflicts=conf)
print("max err", np.abs(X - np.fft.fft(x)).max())
from collections import defaultdict
tot = defaultdict(lambda: [0, 0])
for name, c, i in conf:
    tot[name][0] += c; tot[name][1] += i
for k, v in tot.items(): print(k, "cycles", v[0], "ideal", v[1])
