"""A CPU stand-in for cytvdn_amd.engine.HipBackend -- TEST DOUBLE, lives in tests/ only.

It gives SlabRunner the same five methods (set_params, set_input, step, recon_tensor, sums_tensor)
but performs the slab step with the CPU oracle on CPU torch tensors, so the slab protocol
(SlabLayout bookkeeping + halo exchange over torch.distributed) can be rehearsed with the gloo
backend on a machine without a GPU.  The per-slab semantics it emulates are those documented for
tvdn_iterate_fused in include/tvdn.h: accumulators advance on the own rows and on a high halo row,
recon advances on the own rows, the sums cover the own rows only.
"""
import numpy as np
import torch

from oracle import oracle


class OracleSlabBackend:
    def __init__(self, layout, dtype, fista, max_iters=1):
        self.layout = layout
        self.dtype = np.dtype(dtype)
        self.nd = len(layout.shape)
        tdt = torch.float32 if self.dtype == np.float32 else torch.float64
        ls = layout.local_shape
        self.orig = torch.zeros(ls, dtype=tdt)
        self.recon = torch.zeros(ls, dtype=tdt)
        self.b = [torch.zeros(ls, dtype=tdt) for _ in range(self.nd)]
        self.d = [torch.zeros(ls, dtype=tdt) for _ in range(self.nd)] if fista else None
        self.sums = torch.zeros((max(max_iters, 1), 3), dtype=torch.float64)

    def set_params(self, clip, lam_mu):
        self.clip = np.asarray(clip, self.dtype)
        self.lam_mu = np.asarray(lam_mu, self.dtype)

    def set_input(self, local_block):
        t = local_block if isinstance(local_block, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local_block))
        self.orig.copy_(t)
        self.recon.copy_(t)

    def step(self, tk_ratio, slot):
        lay = self.layout
        own = slice(lay.row_lo, lay.row_hi)
        r = self.recon.numpy()
        bn = 0.0
        for ax in range(self.nd):
            b = self.b[ax].numpy()
            d = self.d[ax].numpy() if tk_ratio is not None else None
            oracle.acc_update(r, b, d, 0.0 if tk_ratio is None else tk_ratio, ax, self.clip[ax], lay.bc_mode)
            bn += float(np.abs(b[own].astype(np.float64)).sum())
        old = r[own].copy()
        oracle.recon_update(self.orig.numpy(), r, [x.numpy() for x in self.b], self.lam_mu, lay.bc_mode)
        self.sums[slot, 0] = bn
        self.sums[slot, 1] = float(np.abs((r[own] - old).astype(np.float64)).sum())
        self.sums[slot, 2] = float(np.abs(old.astype(np.float64)).sum())

    def recon_tensor(self):
        return self.recon

    def sums_tensor(self):
        return self.sums
