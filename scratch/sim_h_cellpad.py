import math, numpy as np
exec(open(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), 'sim_h_layouts.py')).read().split("ident_col =")[0])
def cellpad_cost(S, lane_col, lane_par, cols=32):
    tot=[]; 
    for st in range(25):
        ox0=st*cols
        starts=np.array([n0(ox0+c) for c in range(cols+8)])  # cell starts (absolute px)
        def slot(n):  # n absolute pixel index (array)
            cell=np.searchsorted(starts, n, side='right')-1
            return cell*S + (n-starts[cell])
        col=lane_col[G]; par=lane_par[G]
        base=starts[col]+par                      # [4,16]
        for j in range(TAPS):
            s=slot(base+2*j)                      # [4,16]
            tot.append(conflict(s))
    return float(np.mean(tot))
lanes=np.arange(64)
for S in range(10,40):
    c1=cellpad_cost(S, lanes>>1, lanes&1)
    c2=cellpad_cost(S, lanes&31, lanes>>5)
    print(S, round(c1,3), round(c2,3))
