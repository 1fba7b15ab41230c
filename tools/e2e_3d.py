import json, os, sys, time, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cytvdn_amd as tv
from cytvdn_amd import _lib, synth
shape=(1024,1024,1024)
_lib.ctx(0)
buf=torch.empty(shape,dtype=torch.float32,device="cuda")
_lib.check(_lib.lib().tvdn_synth_fill(0,3,_lib.shape_arr(shape),synth.SEED_3D,0,shape[0],buf.data_ptr(),_lib.current_stream(0)))
x=buf.cpu().numpy(); del buf; torch.cuda.empty_cache()
mu=np.array([1,1,.5],np.float32)
res={}
for pipe in ("1","0","0"):
    os.environ["TVDN_PIPELINE"]=pipe
    t0=time.perf_counter(); r=tv.denoise3D(x,mu,50,FISTA=True,quiet=True); t=time.perf_counter()-t0
    h=hashlib.sha1(r[0].tobytes()).hexdigest()
    print(json.dumps({"pipeline":pipe,"seconds":round(t,3),"Gvoxel_iters_per_s":round(2**30*50/t/1e9,1),"sha":h[:12],"b_norm_last":float(r[1][-1])}),flush=True)
