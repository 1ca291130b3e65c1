import math, itertools, random
GROUPS = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
          list(range(4,12))+list(range(16,20))+list(range(28,32)),
          list(range(32,36))+list(range(44,48))+list(range(52,60)),
          list(range(36,44))+list(range(48,52))+list(range(60,64))]
ratio = 7680/800
support = 2*ratio   # Mitchell, downsampling: radius 2 * ratio
def n0(ox):
    center = (ox + 0.5)*ratio - 0.5
    return int(math.floor(center - support + 0.5))
TAPS = 20
def cycles(w4, perm=None, strips=range(0,25)):
    tot = 0; base = 0
    for st in strips:
        ox0 = st*32
        cols = list(range(32))
        if perm: cols = perm
        cx0 = (n0(ox0)//4)*4
        for j in range(TAPS):
            for g in GROUPS:
                slots = {}
                for lane in g:
                    c = cols[lane>>1]; par = lane&1
                    n = n0(ox0+c) - cx0 + par + 2*j
                    s = (n&3)*w4 + (n>>2)
                    slots.setdefault(s%16, set()).add(s)
                tot += max(len(v) for v in slots.values())
                base += 1
    return tot/base
for k in range(16):
    print(k, round(cycles(1000+ (k - 1000%16)%16 ),3))

def cost_perm(w4, perm, st):
    return cycles(w4, perm, [st])
random.seed(1)
for w4 in (1012-1012%16+4, 1012-1012%16+13):
  for st in range(5):
    perm = list(range(32)); best = cost_perm(w4, perm, st); start = best
    T = 0.05
    for it in range(6000):
        a, b = random.sample(range(32), 2)
        perm[a], perm[b] = perm[b], perm[a]
        c = cost_perm(w4, perm, st)
        if c <= best or random.random() < math.exp((best - c)/T):
            best = c
        else:
            perm[a], perm[b] = perm[b], perm[a]
        T = max(0.002, T*0.999)
    print("w4%16 =", w4 % 16, "strip", st, "identity", round(start,3), "annealed", round(best,3))
