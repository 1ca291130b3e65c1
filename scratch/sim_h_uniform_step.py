import math, random, itertools, numpy as np
ratio = 7680/800; support = 2*ratio
def n0(ox):
    center = (ox + 0.5)*ratio - 0.5
    return int(math.floor(center - support + 0.5))
def residues(st, a, b, cols=32):
    cx0=(n0(st*cols)//4)*4
    out=[]
    for c in range(cols):
        n=n0(st*cols+c)-cx0
        r=[(a*((n+p)>>1)+b*((n+p)&1))%16 for p in (0,1)]
        out.append(tuple(r))
    return out
def partition(res, tries=300):
    """4 groups of 8 columns, 16 residues distinct in each.  returns groups or None"""
    cols=list(range(len(res)))
    for t in range(tries):
        random.shuffle(cols)
        groups=[[] for _ in range(4)]; used=[set() for _ in range(4)]
        ok=True
        for c in cols:
            r=res[c]
            if r[0]==r[1]: ok=False; break
            cand=[g for g in range(4) if len(groups[g])<8 and r[0] not in used[g] and r[1] not in used[g]]
            if not cand: ok=False; break
            g=min(cand,key=lambda g:len(groups[g])) if random.random()<0.5 else random.choice(cand)
            groups[g].append(c); used[g].update(r)
        if ok: return groups
    return None
random.seed(3)
good=[]
for a in range(1,16,2):
    for b in range(16):
        ok=all(partition(residues(st,a,b)) is not None for st in range(25))
        if ok: good.append((a,b))
print("perfect (a,b):", good)
from collections import Counter
for a in (1,3,5,7):
    for b in range(16):
        mx=[]
        for st in range(25):
            cnt=Counter(r for rr in residues(st,a,b) for r in rr)
            mx.append(max(cnt.values()))
        fails=sum(partition(residues(st,a,b),100) is None for st in range(25))
        print(a,b,"max residue count per strip:",max(mx),"mean",sum(mx)/25,"strips without a perfect partition:",fails)
