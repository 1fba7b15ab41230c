import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, cytvdn_amd as tv, sys, time
from cytvdn_amd import synth
for shape in ((128,128,512),(64,64,64,64),(256,256,256)):
    nd=len(shape)
    x=synth.cube(shape,dtype=np.float32); mu=np.array([1,1,.5,.5][:nd],np.float32)
    fn = tv.denoise4D if nd==4 else tv.denoise3D
    fn(x,mu,10,FISTA=True,quiet=True)
    for it in (0,10):
        print("----", shape, it, file=sys.stderr)
        t0=time.perf_counter(); fn(x,mu,it,FISTA=True,quiet=True); print("wall ms", (time.perf_counter()-t0)*1e3, file=sys.stderr)
