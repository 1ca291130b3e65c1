import math, itertools, random, numpy as np, sys
GROUPS = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
          list(range(4,12))+list(range(16,20))+list(range(28,32)),
          list(range(32,36))+list(range(44,48))+list(range(52,60)),
          list(range(36,44))+list(range(48,52))+list(range(60,64))]
G = np.array(GROUPS)  # [4,16] lanes
ratio = 7680/800
support = 2*ratio
def n0(ox):
    center = (ox + 0.5)*ratio - 0.5
    return int(math.floor(center - support + 0.5))
TAPS=20
N0 = np.array([[n0(st*32+c) - (n0(st*32)//4)*4 for c in range(32)] for st in range(25)])  # [25,32]
def read_pixels(lane_col, lane_par):
    """-> array [25*TAPS*4, 16] of pixel indices read together"""
    col = lane_col[G]; par = lane_par[G]              # [4,16]
    base = N0[:, col] + par[None]                      # [25,4,16]
    j = np.arange(TAPS)[None,:,None,None]*2
    px = base[:,None] + j                              # [25,20,4,16]
    return px.reshape(-1,16)
def conflict(slots, nb=16):
    """slots [R,L] -> mean over rows of max distinct addresses per bank"""
    R,L = slots.shape
    bank = slots % nb
    key = bank*100000 + slots
    key.sort(axis=1)
    newaddr = np.ones_like(key, bool); newaddr[:,1:] = key[:,1:] != key[:,:-1]
    b = key//100000
    cnt = np.zeros((R,nb), int)
    np.add.at(cnt, (np.repeat(np.arange(R),L), b.ravel()), newaddr.ravel())
    return cnt.max(axis=1).mean()
ident_col = np.arange(64)>>1; ident_par = np.arange(64)&1
PX = read_pixels(ident_col, ident_par)
def planes_layout(m, beta, alpha=1):
    beta = np.array(beta)
    return lambda n: (n//m)*alpha + beta[n % m]
# write groups: 8 contiguous lanes; lane t holds pixel t+64k (dword mapping): consecutive pixels
WR_dword = np.arange(0,384).reshape(-1,8)
# x4 mapping: lane t holds pixels 4t..4t+3; write q: lanes 8g..8g+7 -> pixels 4*(8g+i)+q
WR_x4 = np.array([[4*(8*g+i)+q for i in range(8)] for g in range(11) for q in range(4)])
def evaluate(layout):
    return conflict(layout(PX)), conflict(layout(WR_dword), 8), conflict(layout(WR_x4), 8)
print("current:", evaluate(planes_layout(4,[0,1012,2024,3036])))
res=[]
for m in (2,4):
    rng = itertools.product(range(16), repeat=m-1)
    for b in rng:
        beta=[0]+[x+16*200*(i+1) for i,x in enumerate(b)]
        r,w1,w4 = evaluate(planes_layout(m,beta))
        res.append((r,w1,w4,m,b))
res.sort()
print(res[:15])
