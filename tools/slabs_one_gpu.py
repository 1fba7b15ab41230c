import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cytvdn_amd as tv
from cytvdn_amd import _lib, synth
shape=(256,256,128,128)
_lib.ctx(0)
buf=torch.empty(shape,dtype=torch.float32,device="cuda")
_lib.check(_lib.lib().tvdn_synth_fill(0,4,_lib.shape_arr(shape),synth.SEED_4D,0,shape[0],buf.data_ptr(),_lib.current_stream(0)))
x=buf.cpu().numpy(); del buf; torch.cuda.empty_cache()
mu=np.array([1,1,.5,.5],np.float32)
os.environ["TVDN_PIPELINE"]="0"
ref=None
for dev in (0,[0,0],[0,0,0,0],[0]*8):
    for rep in range(2):
        t0=time.perf_counter(); r=tv.denoise4D(x,mu,50,quiet=True,device=dev); t=time.perf_counter()-t0
    import hashlib
    h=hashlib.sha1(r[0].tobytes()).hexdigest()
    ref=ref or h
    print(json.dumps({"devices":dev,"seconds":round(t,3),"same_bits":h==ref,"b_norm_last":float(r[1][-1])}),flush=True)
