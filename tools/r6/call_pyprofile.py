import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cProfile, pstats, time
import numpy as np, cytvdn_amd as tv
from cytvdn_amd import synth
shape=tuple(int(v) for v in (sys.argv[1] if len(sys.argv)>1 else "64x64x64x64").split("x"))
fn = tv.denoise4D if len(shape)==4 else tv.denoise3D
x=synth.cube(shape,dtype=np.float32); mu=np.array([1,1,.5,.5][:len(shape)],np.float32)
fn(x,mu,10,FISTA=True,quiet=True)
fn(x,mu,0,FISTA=True,quiet=True)
pr=cProfile.Profile(); pr.enable()
for _ in range(20): fn(x,mu,0,FISTA=True,quiet=True)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
