"""Exhaustive search of the XOR swizzle of the unpadded L = 1024 tile image (fft_tile.h tile_swz): unit index (4 pt + u) ^ mask(pt), mask linear
over GF(2) in bits 1, 2, 3 (all patterns) and 8, 9 (the quarter gathers of the internal layout) of the point index; every access pattern of the
kernel, float and double, is priced with tools/lds_sim.py and only masks at the ideal cycle count survive.  Round 6: (5, 2, 4, 5, 3)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lds_sim import cycles, ideal
import numpy as np, itertools
LOGL=int(sys.argv[1]) if len(sys.argv)>1 else 10
L=1<<LOGL; PP=4; TPT=L//8; WG=TPT*PP
RAD=([1<<(LOGL%3)] if LOGL%3 else [])+[8]*(LOGL//3)
QB=(LOGL-2,LOGL-1)      # the bits of the point index that tell the quarters apart
def mk(m8,m9):
    masks={}; masks[QB[0]]=m8; masks[QB[1]]=m9
    def ua(pt,u):
        x=(pt<<2)|u
        for b,mv in masks.items():
            if (pt>>b)&1: x^=mv
        return x
    return ua
def pat_cost(kind, fn):
    tot=0
    for w in range(0,WG,64):
        tot+=cycles(kind,[fn(w+l)*16 for l in range(64)])
    return tot/(WG//64)/ideal(kind)
def patterns(ua,S):
    out={}
    # P2 reads
    out['read']=pat_cost('r128', lambda tid: ua(tid//4+TPT*3, tid%4))
    # stage writes
    Ns=1
    for s,R in enumerate(RAD):
        for d in range(R):
            def f(tid,d=d,Ns=Ns,R=R):
                t=tid//4; p=tid%4; tk=t%Ns; tq=t//Ns
                return ua((tq+1*(TPT//Ns))*(Ns*R)%L+tk+d*Ns if False else (tq*(Ns*R)+tk+d*Ns)%L, p)
            out[f'w{s}d{d}']=pat_cost('w128', f)
        Ns*=R
    out['final']=pat_cost('r128', lambda tid: ua((tid+WG*3)//4, tid%4))
    if S==2:
        for h in (0,1):
            for u in range(4):
                out[f'tw{h}{u}']=pat_cost('w128', lambda tid: ua(2*tid+h, u))
        for k in (0,1):
            def f(tid,k=k):
                g=tid+WG*1; ptq=g//16; r=g%16; bb=r//8; m=(r//2)%4
                return ua(ptq+(L//4)*m, 2*bb+k)
            out[f'oint{k}']=pat_cost('r128', f)
        def f(tid):
            g=tid+WG*1; ptq=g//16; r=g%16; bb=r//8; m=(r//2)%4; sub=r%2
            return ua(ptq+(L//4)*m, 2*bb+sub)
        out['iint']=pat_cost('w128', f)
    else:
        for i in range(2):
            for sq in range(4):
                out[f'tw{i}{sq}']=pat_cost('w128', lambda tid: ua(tid+WG*i, sq))
        for k in (0,1):
            def f(tid,k=k):
                g=tid+WG*1; ptq=g//16; r=g%16; m=(r//4)%4; sub=r%4
                return ua(ptq+(L//4)*m, 2*(sub&1)+k)
            out[f'oint{k}']=pat_cost('r128', f)
        def f(tid):
            g=tid+WG*1; ptq=g//16; r=g%16; m=(r//4)%4; sub=r%4
            return ua(ptq+(L//4)*m, 2*(sub&1)+(sub>>1))
        out['iint']=pat_cost('w128', f)
    return out
def mk2(masks):
    def ua(pt,u):
        x=(pt<<2)|u
        for b,mv in masks.items():
            if (pt>>b)&1: x^=mv
        return x
    return ua
import itertools
def base_patterns(ua,S):
    o=patterns(ua,S)
    return {k:v for k,v in o.items() if not k.startswith('oint') and k!='iint'}
res=[]
for m1,m2,m3 in itertools.product(range(16),repeat=3):
    masks={1:m1,2:m2,3:m3}
    ua=mk2(masks)
    seen=set(ua(pt,u) for pt in range(64) for u in range(4))
    if len(seen)!=256 or max(seen)>=256: continue
    c2=base_patterns(ua,2)
    if max(c2.values())>1: continue
    c1=base_patterns(ua,1)
    if max(c1.values())>1: continue
    res.append((m1,m2,m3))
print(len(res), res[:20])
found=[]
for (m1,m2,m3) in res:
    for m8 in range(16):
        for m9 in range(16):
            masks={1:m1,2:m2,3:m3,QB[0]:m8,QB[1]:m9}
            ua=mk2(masks)
            seen=set(ua(pt,u) for pt in range(L) for u in range(4))
            if len(seen)!=4*L or max(seen)>=4*L: continue
            c2=patterns(ua,2); c1=patterns(ua,1)
            w=max(max(c2.values()),max(c1.values()))
            t=sum(c2.values())+sum(c1.values())
            found.append((w,t,(m1,m2,m3,m8,m9),{k:v for k,v in c2.items() if v>1},{k:v for k,v in c1.items() if v>1}))
    if len(found)>4000: break
found.sort(key=lambda x:(x[0],x[1]))
for f in found[:10]: print(f)
print([f[2] for f in found if f[0]==1.0][:20])
print("masks below 8 only:", [f[2] for f in found if f[0]==1.0 and max(f[2])<8][:10])
