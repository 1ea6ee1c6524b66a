import numpy as np, sys
sys.path.insert(0,'tools')
import lds_sim
L = np.arange(64)
k1p, c = L>>3, L&7
best=[]
for S2 in range(128, 200, 2):
  for RS in (8, 10, 12, 16, 18):
    if 16*RS > S2: continue
    for sw in (0,1,2,3):
      def A(k1,ka,cc):
        # optional xor swizzle of pair index by ka / k1 bits
        pair = cc>>1
        if sw==1: pair = pair ^ (ka & 3)
        if sw==2: pair = pair ^ (k1 & 3)
        if sw==3: pair = pair ^ ((ka>>2) & 3)
        return k1*S2 + ka*RS + (pair<<1) + (cc&1)
      w = sum(lds_sim.cycles("w64", A(k1p, ka, c)*8) for ka in range(16))
      r = 0
      for e in range(2):
        k1 = 2*(L&3)+e; ka = L>>2
        for c0 in range(0,8,2):
          r += lds_sim.cycles("r128", A(k1,ka,c0)*8)
      best.append((w+r, w, r, S2, RS, sw))
best.sort()
for b in best[:12]: print(b)
