import math, random, numpy as np, sys
exec(open(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), 'sim_h_layouts.py')).read().split("ident_col =")[0])
def cost(perm, beta, m, halves):
    lanes = np.arange(64)
    if halves:
        col = perm[lanes & 31]; par = lanes >> 5
    else:
        col = perm[lanes >> 1]; par = lanes & 1
    px = read_pixels(col, par)
    b = np.array(beta)
    return conflict((px//m) + b[px % m])
random.seed(int(sys.argv[1]) if len(sys.argv)>1 else 1)
for m in (4,8):
  for halves in (0,1):
    perm = np.arange(32); beta=[0]+[random.randrange(16)+16*300*(i+1) for i in range(m-1)]
    best = cost(perm,beta,m,halves); T=0.05
    for it in range(4000):
        if random.random()<0.5:
            a,b_=random.sample(range(32),2); perm[a],perm[b_]=perm[b_],perm[a]
            c=cost(perm,beta,m,halves)
            if c<=best or random.random()<math.exp((best-c)/T): best=c
            else: perm[a],perm[b_]=perm[b_],perm[a]
        else:
            i=random.randrange(1,m); old=beta[i]; beta[i]=random.randrange(16)+16*300*i
            c=cost(perm,beta,m,halves)
            if c<=best or random.random()<math.exp((best-c)/T): best=c
            else: beta[i]=old
        T=max(0.002,T*0.9985)
    print("m",m,"halves",halves,"best",round(best,3),"beta",[x%16 for x in beta],"perm",perm.tolist(),flush=True)
