import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, cytvdn_amd as tv, json
from cytvdn_amd import synth
for shape in ((128,128,512),(128,128,513),(128,128,510),(127,129,512),(64,64,124,124),(64,64,125,125),(64,64,126,126),(63,65,124,124)):
    nd=len(shape)
    x=synth.cube(shape,dtype=np.float32); mu=np.array([1,1,.5,.5][:nd],np.float32)
    fn = tv.denoise4D if nd==4 else tv.denoise3D
    fn(x,mu,5,FISTA=True,quiet=True)
    t=[]
    for it in (20,120):
        t0=time.perf_counter(); fn(x,mu,it,FISTA=True,quiet=True); t.append(time.perf_counter()-t0)
    per=(t[1]-t[0])/100
    print(json.dumps({"shape":shape,"us_per_iter":round(per*1e6,1),"Gvoxel_iters_per_s":round(np.prod(shape)/per/1e9,2)}),flush=True)
