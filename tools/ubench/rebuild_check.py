import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from cytvdn_amd import _lib, synth
from cytvdn_amd.engine import HipBackend, SlabLayout, fista_ratios
for shape, dt, nf, npl in (((23,3,4,8), np.float32, 6, 0), ((12,6,16), np.float64, 5, 0), ((9,2,5,7), np.float32, 4, 3), ((23,3,4,8), np.float32, 0, 5), ((16,4,8,16), np.float32, 7, 0)):
    nd=len(shape); dt=np.dtype(dt)
    x = synth.cube(shape, seed=61, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt); lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    be = HipBackend(SlabLayout(shape, 0, 1, 2), dt, nf > 0, device=0, max_iters=nf+npl)
    be.set_params(1.0/lam, (lam/mu).astype(dt)); be.set_input(x)
    r = fista_ratios(nf)
    for i in range(nf): be.step(float(r[i]), i)
    for i in range(npl): be.step(None, nf+i)
    torch.cuda.synchronize()
    cur = be.recon[be.cur].clone()
    out = torch.full_like(cur, 7.0)
    P = C.c_void_p * nd
    if be.d_form:
        d = P(*[be.S[q][be.i_d].data_ptr() for q in range(nd)]); dp = P(*[be.S[q][be.i_prev].data_ptr() for q in range(nd)])
    else:
        d = P(*[be.S[q][be.i_b].data_ptr() for q in range(nd)]); dp = None
    lm = (C.c_double * nd)(*[float(v) for v in (lam/mu).astype(dt)])
    _lib.check(_lib.lib().tvdn_recon_from_state(be.code, nd, _lib.shape_arr(shape), be.orig.data_ptr(), out.data_ptr(), d, dp, lm, float(be.tk_prev), 0, shape[0], None))
    torch.cuda.synchronize()
    same = torch.equal(out.view(torch.int32 if dt==np.float32 else torch.int64), cur.view(torch.int32 if dt==np.float32 else torch.int64))
    print(shape, dt.name, nf, npl, 'd_form', be.d_form, 'bit-identical', same, 'max abs diff', float((out-cur).abs().max()))
