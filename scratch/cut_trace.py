import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, timg_amd, oracle_lib
from timg_amd import synth
o=oracle_lib.Oracle()
fb=o.scale(synth.make("photo",3840,2160,seed=0),800,450)
hip=timg_amd.TimgHip(0)
for _ in range(2):
    out=hip.sixel_encode(fb,800,450)[0]
print(len(out))
