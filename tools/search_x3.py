import numpy as np, sys
sys.path.insert(0,'tools')
import lds_sim
L = np.arange(64)
res=[]
for P in range(1024, 1024+200, 4):
  for MP in (0,4,8,12,16,20,24,32):
    for BP in (0,4):   # pad per 32 k (per b-group of 8)
      def PA(p,k): return p*P + k + MP*(k>>8) + BP*((k>>5)&7)
      if PA(0,1023)+1 > P: continue
      w = 0
      for kc in range(8):
        k = 2*L + 128*kc
        for p in range(2):
          w += lds_sim.cycles("w64", PA(p,k)*4)
      r = 0
      for s in range(8):
        b = 8*s + (L>>3); m = (L>>1)&3; p = L&1
        r += lds_sim.cycles("r128", PA(p, 256*m+4*b)*4)
      res.append((w+r, w, r, P, MP, BP))
res.sort()
for x in res[:10]: print(x)
print("ideal w", 16*4, "r", 8*4)
