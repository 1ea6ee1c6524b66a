import numpy as np, sys
sys.path.insert(0,'tools')
import lds_sim
L = np.arange(64)
res=[]
for P in range(1024, 1024+260, 4):
  for MP in (0,4,8,12,16,20,24,32,36,40,48):
    for BP in (0,4,8):
      def PA(p,k): return p*P + k + MP*(k>>8) + BP*((k>>5)&7)
      if PA(0,1023)+4 > P: continue
      w = 0
      for s in range(8):
        b = 8*s + (L>>3); m = (L>>1)&3; p = L&1
        w += lds_sim.cycles("w128", PA(p, 256*m+4*b)*4)
      r = 0
      for j in range(8):
        k = 128*j + 2*L
        for p in range(2):
          r += lds_sim.cycles("r64", PA(p,k)*4)
      # also X3 direction (w64 / r128) must stay good
      w3 = 0
      for kc in range(8):
        k = 2*L + 128*kc
        for p in range(2): w3 += lds_sim.cycles("w64", PA(p,k)*4)
      r3 = 0
      for s in range(8):
        b = 8*s + (L>>3); m = (L>>1)&3; p = L&1
        r3 += lds_sim.cycles("r128", PA(p, 256*m+4*b)*4)
      res.append((w+r+w3+r3, w, r, w3, r3, P, MP, BP))
res.sort()
for x in res[:8]: print(x)
print("ideal w128", 8*8, "r64", 16*2, "w64", 16*4, "r128", 8*4)
cur=[x for x in res if x[5]==1120 and x[6]==16 and x[7]==0]; print("current", cur)
