"""Out-of-core engine, wavefront schedule: k iterations per PCIe round trip with NO redundant sweeps.

`outofcore.StagedRunner` advances a block k iterations by letting the swept range shrink one row per
iteration on each interior side (a trapezoid in row x iteration space): simple, but at k ~ block height half
of the sweeps are spent on halo rows.  Here the blocks are parallelograms instead: the cube streams
through the GPU once per pass, in chunks of R rows, and iteration level j+1 trails level j by one row --
chunk c computes, for j = 0..k-1, rows [cR-(j+1), (c+1)R-(j+1)) of level j+1 from level j -- so every row of
every level is computed exactly once and crosses PCIe once per k iterations (10 arrays up, 9 down).

Each level keeps a sliding window of R+3 rows per state array in HBM (recon_j and one accumulator array per
axis; in the compact FISTA state level j's `d_j` also serves as `d_prev` of level j+1's update), plus one
window of the input.  The sweeps are the same `tvdn_iterate_fused` launches as everywhere else: windows are
presented to the kernel as row ranges of virtual arrays by offsetting the base pointers, so the arithmetic
-- and the bits -- are those of the in-core engine.  Jia-Zhao and (single process) periodic BC; slabs across
ranks; `reference_data` traces; no per-iteration host decisions (a stopping rule makes `driver._run_staged`
fall back to the trapezoid engine with k = 1).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .engine import fista_ratios


def _write_or_copy(own: np.ndarray, out):
    if out is None:
        return own.copy()
    row_bytes = max(1, own[0].nbytes)
    step = max(1, min(own.shape[0], (256 << 20) // row_bytes))
    for a in range(0, own.shape[0], step):
        out.write_rows(a, own[a:a + step])
    return None


IO_STREAMS = 1      # HIP streams per PCIe direction of the wavefront engine (TVDN_IO_STREAMS overrides)
WINDOW_SLACK = 2    # rows a level window holds beyond the chunk height (planner.wavefront_windows uses the same)


class _Window:
    """Sliding window of rows [base, top) of one array, stored from buffer row 0."""

    def __init__(self, cap, plane, tdt, dev):
        self.buf = torch.zeros((cap,) + tuple(plane), dtype=tdt, device=dev)
        self.base = 0
        self.top = 0

    def slide(self, new_base, pend=None):
        """Drop the rows below new_base (global index), keeping [new_base, top) at the front.  With `pend` (a list) a
        non-overlapping move is not performed but appended as a (dst, src) pair for one batched launch
        (_lib.copy_many): per chunk some 150 windows slide, and as individual runtime copies they were 81 % of a pass."""
        if new_base <= self.base:
            return
        keep = max(0, self.top - new_base)
        shift = new_base - self.base
        if keep > 0:
            src = self.buf[shift:shift + keep]
            if pend is not None and shift >= keep:
                pend.append((self.buf[:keep], src))
            else:
                self.buf[:keep].copy_(src.clone() if shift < keep else src)
        self.base = new_base
        self.top = max(self.top, new_base)

    def rows(self, g0, g1):
        """View of global rows [g0, g1)."""
        assert self.base <= g0 and g1 - self.base <= self.buf.shape[0], (self.base, g0, g1, self.buf.shape[0])
        return self.buf[g0 - self.base:g1 - self.base]

    def ptr(self, ref, row_bytes):
        """Base pointer of a virtual array whose row 0 is global row `ref` (may lie outside the buffer;
        only rows inside the window are ever dereferenced)."""
        return self.buf.data_ptr() + (ref - self.base) * row_bytes


class WavefrontRunner:
    """Host-resident cube, k iterations per streaming pass, no redundant sweeps (see module docstring)."""

    def __init__(self, datacube: np.ndarray, fista: bool, clip, lam_mu, device: int = 0, chunk_rows: int = 16,
                 k: int = 32, max_iters: int = 1, pin: bool = True, global_rows: int = None, row0: int = 0,
                 group=None, world: int = 1, rank: int = 0, bc_mode: int = 2, reference: np.ndarray = None,
                 exact_wrap: bool = False, host_inplace: bool = True):
        """Slab mode (`world` > 1): `datacube` holds this rank's own rows [row0, row0+rows) of a cube with
        `global_rows` rows.  The host arrays then carry up to k extra rows per interior side, refreshed from the
        neighbouring ranks before every pass; at those artificial faces the wavefront gives up one row per
        level (a trapezoid k rows wide), everywhere else it stays redundancy-free."""
        own_shape = tuple(int(s) for s in datacube.shape)
        if bc_mode not in (0, 2):
            raise NotImplementedError("BC_mode must be 0 (periodic) or 2 (Jia-Zhao)")
        if bc_mode == 0 and world > 1:
            raise NotImplementedError("periodic BC across staged slabs is not built; use the in-core slab engine")
        # Periodic BC along axis 0 (single process): the cube is extended by k wrapped rows at both ends, which
        # makes both faces "artificial" exactly like slab faces whose neighbour is the cube's own other end.
        self.bc = int(bc_mode)
        self.periodic = (self.bc == 0)
        self.nd = len(own_shape)
        self.dtype = datacube.dtype
        self.code = _lib.dtype_code(self.dtype)
        self.fista = bool(fista)
        self.device = int(device)
        self.k = max(1, int(k))
        self.R = max(2, int(chunk_rows))
        if self.periodic:
            k = min(int(k), own_shape[0])
        self.k = max(1, int(k))
        self.N0 = int(own_shape[0] if global_rows is None else global_rows) + (2 * self.k if self.periodic else 0)
        self.g0 = int(row0) + (self.k if self.periodic else 0)
        self.g1 = self.g0 + own_shape[0]
        self.world, self.rank, self.group = int(world), int(rank), group
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            if own_shape[0] < self.k:
                raise ValueError(f"a slab needs at least k = {self.k} rows (it has {own_shape[0]})")
        halo = self.world > 1 or self.periodic
        self.ext_lo = min(self.k, self.g0) if halo else 0
        self.ext_hi = min(self.k, self.N0 - self.g1) if halo else 0
        self.base = self.g0 - self.ext_lo                    # global index of host row 0
        self.shape = (self.ext_lo + own_shape[0] + self.ext_hi,) + own_shape[1:]
        self.max_iters = max(1, int(max_iters))
        self.ctx = _lib.ctx(self.device)
        tdt = torch.float32 if self.dtype == np.float32 else torch.float64
        self.tdt = tdt
        dev = torch.device("cuda", self.device)
        self.dev = dev
        plane = self.shape[1:]
        self.row_bytes = int(np.prod(plane)) * self.dtype.itemsize
        self.clip = [float(v) for v in clip]
        self.lam_mu = [float(v) for v in lam_mu]

        own_sl = slice(self.ext_lo, self.ext_lo + own_shape[0])

        def host(fill=None):
            t = torch.empty(self.shape, dtype=tdt, pin_memory=pin)
            t.zero_()
            if fill is not None and hasattr(fill, "read_rows"):      # a cube on disk (cubeio.LazyCube): block by block
                own = t[own_sl]
                step = fill.block_rows()
                for a in range(0, own.shape[0], step):
                    own[a:a + step].copy_(torch.from_numpy(fill.read_rows(a, min(a + step, own.shape[0]))))
            elif fill is not None:
                t[own_sl].copy_(torch.from_numpy(fill))
            return t

        def as_source(x):
            return x if hasattr(x, "read_rows") else np.ascontiguousarray(x)

        # Host state.  A pass reads row r of the old state (upload) at least k rows ahead of where it writes the new
        # state (download), so old and new can be the SAME pinned arrays: 1 + 1 + nd x n_state of them (10 for 4-D
        # FISTA) instead of 19 -- 2.5 TiB instead of 4.9 TiB for BASELINE config 5, which is what fits a 3 TB host.
        # `host_inplace=False` keeps separate old/new arrays (cross-check).
        self.orig_h = host(as_source(datacube))
        n_state = 2 if self.fista else 1
        r0 = host(as_source(datacube))
        s0 = [[host() for _ in range(n_state)] for _ in range(self.nd)]
        if host_inplace:
            self.recon_h, self.state_h = [r0, r0], [s0, s0]
        else:
            self.recon_h = [r0, host()]
            self.state_h = [s0, [[host() for _ in range(n_state)] for _ in range(self.nd)]]
        self.ref_h = host(np.ascontiguousarray(reference)) if reference is not None else None
        self.mse_dev = torch.zeros(self.max_iters + 1, dtype=torch.float64, device=dev) if reference is not None else None
        self._sse_tmp = torch.zeros(1, dtype=torch.float64, device=dev)
        self.h_old = 0
        self.d_form = self.fista
        self.tk_prev = 0.0
        self.iters_done = 0
        self.sums_dev = torch.zeros((self.max_iters + 1, 3), dtype=torch.float64, device=dev)   # last row: discard slot
        self._swap = None
        if self.world > 1:
            from .outofcore import HaloSwap
            # the neighbours' rows of the state: the low halo rows at once, the high ones while the pass streams upward
            self._swap = HaloSwap(self.dist, self.group, self.rank, self.world, self.device)
            self._exchange = lambda arrays, depth, overlap=False: self._swap.start(
                arrays, self.ext_lo, self.ext_lo + own_shape[0], self.ext_lo, self.ext_hi, depth, overlap=overlap)
            self._exchange([self.orig_h], self.k)
        elif self.periodic:
            self._exchange = self._wrap_rows
            self._exchange([self.orig_h], self.k)
        self.bytes_h2d = 0
        self.bytes_d2h = 0

        # a launch that brings rows [a, b) to level j+1 reads rows a-1 .. b of level j: R + 2 rows per window (round 1
        # kept R + 3; one row less per window is 20 % less HBM at 2-row chunks, i.e. a deeper k for the same memory)
        cap = self.R + WINDOW_SLACK
        K = self.k
        # levels -1 .. K: recon (levels 0..K) and one accumulator array per axis (levels -1..K)
        self.Rw = [_Window(cap, plane, tdt, dev) for _ in range(K + 1)]
        self.Aw = [[_Window(cap, plane, tdt, dev) for _ in range(self.nd)] for _ in range(K + 2)]  # index level + 1
        self.Ow = _Window(self.R + K + 3, plane, tdt, dev)
        self.Fw = _Window(self.R + K + 3, plane, tdt, dev) if reference is not None else None   # reference_data rows
        n_in = 3 + 2 * self.nd
        n_out = 1 + 2 * self.nd
        self.inbox = [[torch.empty((self.R,) + tuple(plane), dtype=tdt, device=dev) for _ in range(n_in)] for _ in range(2)]
        self.outbox = [[torch.empty((self.R,) + tuple(plane), dtype=tdt, device=dev) for _ in range(n_out)] for _ in range(2)]
        # HIP streams per PCIe direction.  One each is best: up and down together already hold the link at 32 + 29 GB/s
        # (256 MiB planes, 32x1024x256x256, 2 rows x k = 24: 19.1 Gvoxel-iters/s with 1 stream per direction, 18.4
        # with 2, 13.9 with 4); the knob stays for other hosts.
        n_io = max(1, int(os.environ.get("TVDN_IO_STREAMS", str(IO_STREAMS))))
        self.ups = [torch.cuda.Stream(device=dev) for _ in range(n_io)]
        self.downs = [torch.cuda.Stream(device=dev) for _ in range(n_io)]
        self._args = _lib.IterArgs()
        # Jia-Zhao, `exact_wrap` (single process): the sweeps at the cube's top face form the wrapped axis-0 accumulator
        # from the recon of global row 0 AT THEIR OWN LEVEL (TVDN_EDGE_WRAP) instead of taking it as zero, which it is
        # only while row 0 is finite (engine.py, "Non-finite data").  Row 0 of every level is computed at the start of a
        # pass and has long left its window when the top is reached, so one plane per level is kept aside.
        self.row0 = None
        if exact_wrap and self.world == 1 and not self.periodic:
            self.row0 = [torch.empty(tuple(plane), dtype=tdt, device=dev) for _ in range(K + 1)]

    def _wrap_rows(self, arrays, depth):
        """Periodic BC: the halo rows below the first / above the last own row are the cube's own other end."""
        lo, hi = self.ext_lo, self.ext_lo + (self.g1 - self.g0)
        for t in arrays:
            t[lo - depth:lo].copy_(t[hi - depth:hi])
            t[hi:hi + depth].copy_(t[lo:lo + depth])

    def _sse(self, a: torch.Tensor, b: torch.Tensor, slot: int):
        """mse[slot] += sum((a - b)^2) over two equally shaped row blocks (sum_square_error, utils.pyx:14-49)."""
        _lib.check(_lib.lib().tvdn_sum_square_error(self.ctx, self.code, self.nd, _lib.shape_arr(a.shape), a.data_ptr(),
                                                    b.data_ptr(), self._sse_tmp.data_ptr(), _lib.current_stream(self.device)))
        self.mse_dev[slot:slot + 1] += self._sse_tmp

    def device_bytes(self) -> int:
        n = (len(self.Rw) + len(self.Aw) * self.nd) * self.Rw[0].buf.numel() + self.Ow.buf.numel()
        n += sum(t.numel() for b in self.inbox + self.outbox for t in b)
        return n * self.dtype.itemsize

    # ---- one launch: level j -> j+1 for global rows [a, b) ---------------------------------------------------
    def _launch(self, j, a, b, tk, tk_prev, mode, slot):
        N0, rb = self.N0, self.row_bytes
        ref = 0 if a == 0 else a - 1
        at_top = (b == N0)
        row_hi = (b - ref) if at_top else (b - ref + 1)
        A = self._args
        A.dtype, A.ndim = self.code, self.nd
        A.shape[0] = row_hi
        for i, s in enumerate(self.shape[1:]):
            A.shape[i + 1] = s
        A.row_lo, A.row_hi = 0, row_hi
        A.sweep_lo, A.sweep_hi = a - ref, b - ref
        A.lo_mode = _lib.EDGE_BC
        A.hi_mode = _lib.EDGE_ZERO if at_top else _lib.EDGE_BC
        A.wrap_recon = None
        if at_top and self.row0 is not None:
            A.hi_mode, A.wrap_recon = _lib.EDGE_WRAP, self.row0[j].data_ptr()
        A.bc_mode = self.bc
        A.mode = mode
        A.tk, A.tk_prev = float(tk or 0.0), float(tk_prev)
        A.accumulate = 1
        A.orig = self.Ow.ptr(ref, rb)
        A.recon_in = self.Rw[j].ptr(ref, rb)
        A.recon_out = self.Rw[j + 1].ptr(ref, rb)
        for q in range(self.nd):
            cur, prv, nxt = self.Aw[j + 1][q], self.Aw[j][q], self.Aw[j + 2][q]
            A.b_in[q] = A.b_out[q] = A.d_in[q] = A.d_out[q] = A.dprev_in[q] = None
            A.clip[q], A.lambda_mu[q] = self.clip[q], self.lam_mu[q]
            if mode == _lib.ITER_FISTA_D:
                A.d_in[q], A.dprev_in[q], A.d_out[q] = cur.ptr(ref, rb), prv.ptr(ref, rb), nxt.ptr(ref, rb)
            elif mode == _lib.ITER_FISTA_D_TO_PLAIN:
                A.d_in[q], A.dprev_in[q], A.b_out[q] = cur.ptr(ref, rb), prv.ptr(ref, rb), nxt.ptr(ref, rb)
            else:
                A.b_in[q], A.b_out[q] = cur.ptr(ref, rb), nxt.ptr(ref, rb)
        _lib.check(_lib.lib().tvdn_iterate_fused(self.ctx, C.byref(A), C.c_void_p(self.sums_dev[slot].data_ptr()),
                                                 _lib.current_stream(self.device)))

    # ---- one pass of len(ratios) iterations over the whole cube ------------------------------------------------
    def _pass(self, ratios, slot0):
        kk, R, N0, nd = len(ratios), self.R, self.N0, self.nd
        old, new = self.h_old, self.h_old ^ 1
        main = torch.cuda.current_stream(self.dev)
        hb = self.base                                      # host row index = global row - hb
        g0, g1 = self.g0, self.g1
        # rows this pass works on: own rows plus kk rows of the neighbours' state at each artificial face
        halo = self.world > 1 or self.periodic
        E0, E1 = max(0, g0 - kk) if halo else 0, min(N0, g1 + kk) if halo else N0
        art_lo = (E0 > 0) or self.periodic                  # faces that are not the cube's own boundary
        art_hi = (E1 < N0) or self.periodic
        if halo:
            arrays = [self.recon_h[old]] + [t for q in range(nd) for t in self.state_h[old][q][: (2 if self.d_form else 1)]]
            if self._swap is not None:
                self._exchange(arrays, kk, overlap=True)     # high halo rows arrive while the pass works its way up
            else:
                self._exchange(arrays, kk)
        discard = self.max_iters
        # form and mode of every level of this pass
        forms = [self.d_form]
        modes, tkp = [], []
        prev_ratio = self.tk_prev
        for j, tk in enumerate(ratios):
            if tk is not None:
                if not forms[j]:
                    raise ValueError("a FISTA iteration cannot follow an unaccelerated one")
                modes.append(_lib.ITER_FISTA_D)
                forms.append(True)
            else:
                modes.append(_lib.ITER_FISTA_D_TO_PLAIN if forms[j] else _lib.ITER_PLAIN)
                forms.append(False)
            tkp.append(prev_ratio)
            if tk is not None:
                prev_ratio = tk
        n_in_state = 2 if forms[0] else 1
        n_out_state = 2 if forms[kk] else 1
        for w in self.Rw[:kk + 1] + [x for lvl in self.Aw[:kk + 2] for x in lvl] + [self.Ow] + ([self.Fw] if self.Fw else []):
            w.base = w.top = 0
        n_chunks = (E1 - E0 + kk + R - 1) // R

        def lo_bound(level):    # lowest row that can be brought to `level` (an artificial face loses a row per level)
            return E0 + level if art_lo else 0

        def hi_bound(level):
            return E1 - level if art_hi else N0
        in_ready, in_free = [None, None], [None, None]
        out_ready, out_free = [None, None], [None, None]

        def upload(c):
            u0, u1 = E0 + c * R, min(E0 + (c + 1) * R, E1)
            if u0 >= u1:
                return
            if self._swap is not None and u1 > g1:
                self._swap.finish()                          # this chunk reads the neighbour's rows above my slab
            box = self.inbox[c % 2]
            n = u1 - u0
            pairs = [(box[0], self.orig_h), (box[1], self.recon_h[old])]
            i = 2
            for q in range(nd):
                for s in range(n_in_state):
                    pairs.append((box[i], self.state_h[old][q][s]))
                    i += 1
            if self.ref_h is not None:
                pairs.append((box[-1], self.ref_h))
            evs = []
            for si, st in enumerate(self.ups):
                with torch.cuda.stream(st):
                    if in_free[c % 2] is not None:
                        st.wait_event(in_free[c % 2])
                    for dst, src in pairs[si::len(self.ups)]:
                        dst[:n].copy_(src[u0 - hb:u1 - hb], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(st)
                    evs.append(ev)
            self.bytes_h2d += len(pairs) * n * self.row_bytes
            in_ready[c % 2] = evs

        upload(0)
        for c in range(n_chunks):
            upload(c + 1)                                   # next chunk crosses PCIe while this one is swept
            u0, u1 = E0 + c * R, min(E0 + (c + 1) * R, E1)
            # slide every window to what chunk c still needs: level j keeps rows >= E0 + cR - j - 2
            pend = []
            for j in range(-1, kk + 1):
                nb = max(0, E0 + c * R - j - 2)
                if j >= 0:
                    self.Rw[j].slide(nb, pend)
                for q in range(nd):
                    self.Aw[j + 1][q].slide(nb, pend)
            self.Ow.slide(max(0, E0 + c * R - kk - 1), pend)
            if self.Fw is not None:
                self.Fw.slide(max(0, E0 + c * R - kk - 1), pend)
            _lib.copy_many(pend, self.device)
            if u0 < u1:
                n = u1 - u0
                box = self.inbox[c % 2]
                for ev in in_ready[c % 2]:
                    main.wait_event(ev)
                pairs = [(self.Ow.rows(u0, u1), box[0][:n]), (self.Rw[0].rows(u0, u1), box[1][:n])]
                i = 2
                for q in range(nd):
                    pairs.append((self.Aw[1][q].rows(u0, u1), box[i][:n]))          # level 0: d_k (or b)
                    i += 1
                    if n_in_state == 2:
                        pairs.append((self.Aw[0][q].rows(u0, u1), box[i][:n]))      # level -1: d_k-1
                        i += 1
                if self.Fw is not None:
                    pairs.append((self.Fw.rows(u0, u1), box[-1][:n]))
                _lib.copy_many(pairs, self.device)
                if self.row0 is not None and u0 == 0:
                    self.row0[0].copy_(self.Rw[0].rows(0, 1)[0])
                if self.Fw is not None:
                    self.Fw.top = u1
                    if self.iters_done == 0:   # MSE[0]: the input against the reference (cyTVDN.py:124-125), own rows
                        o0, o1 = max(u0, g0), min(u1, g1)
                        if o0 < o1:
                            self._sse(self.Rw[0].rows(o0, o1), self.Fw.rows(o0, o1), 0)
                ev = torch.cuda.Event()
                ev.record(main)
                in_free[c % 2] = ev
                self.Ow.top = self.Rw[0].top = u1
                for q in range(nd):
                    self.Aw[1][q].top = self.Aw[0][q].top = u1
            # the wavefront: level j+1 trails level j by one row
            for j in range(kk):
                a = max(lo_bound(j + 1), E0 + c * R - (j + 1))
                b = min(hi_bound(j + 1), E0 + (c + 1) * R - (j + 1))
                if a >= b:
                    continue
                # the sums count own rows only: rows of the neighbours' halo go to a discard slot
                for x0, x1, slot in ((a, min(b, g0), discard), (max(a, g0), min(b, g1), slot0 + j), (max(a, g1), b, discard)):
                    if x0 < x1:
                        self._launch(j, x0, x1, ratios[j], tkp[j], modes[j], slot)
                        if self.row0 is not None and x0 == 0:
                            self.row0[j + 1].copy_(self.Rw[j + 1].rows(0, 1)[0])
                        if self.Fw is not None and slot != discard:
                            self._sse(self.Fw.rows(x0, x1), self.Rw[j + 1].rows(x0, x1), slot + 1)
                self.Rw[j + 1].top = b
                for q in range(nd):
                    self.Aw[j + 2][q].top = b
            # own rows that have reached the last level go home
            a = max(g0, E0 + c * R - kk)
            b = min(g1, E0 + (c + 1) * R - kk)
            if a < b:
                n = b - a
                box = self.outbox[c % 2]
                for ev in out_free[c % 2] or ():
                    main.wait_event(ev)
                pairs = [(box[0][:n], self.Rw[kk].rows(a, b))]
                i = 1
                for q in range(nd):
                    pairs.append((box[i][:n], self.Aw[kk + 1][q].rows(a, b)))
                    i += 1
                    if n_out_state == 2:
                        pairs.append((box[i][:n], self.Aw[kk][q].rows(a, b)))
                        i += 1
                _lib.copy_many(pairs, self.device)
                ev = torch.cuda.Event()
                ev.record(main)
                pairs = [(self.recon_h[new], box[0])]
                i = 1
                for q in range(nd):
                    for s in range(n_out_state):
                        pairs.append((self.state_h[new][q][s], box[i]))
                        i += 1
                evs = []
                for si, st in enumerate(self.downs):
                    with torch.cuda.stream(st):
                        st.wait_event(ev)
                        for dst, src in pairs[si::len(self.downs)]:
                            dst[a - hb:b - hb].copy_(src[:n], non_blocking=True)
                        ev2 = torch.cuda.Event()
                        ev2.record(st)
                        evs.append(ev2)
                self.bytes_d2h += len(pairs) * n * self.row_bytes
                out_free[c % 2] = evs
        for st in self.downs:
            st.synchronize()
        main.synchronize()
        if self._swap is not None:
            self._swap.finish()
        self.h_old = new
        self.d_form = forms[kk]
        self.tk_prev = prev_ratio
        self.iters_done += kk

    def run(self, n_fista: int, n_plain: int):
        slot = self.iters_done
        ratios = [float(r) for r in fista_ratios(n_fista)] + [None] * int(n_plain)
        i = 0
        while i < len(ratios):
            grp = ratios[i:i + self.k]
            self._pass(grp, slot + i)
            i += len(grp)

    def sums(self) -> np.ndarray:
        """[max_iters, 3] f64 sums over the own rows (all-reduced over the ranks in slab mode)."""
        t = self.sums_dev[: self.max_iters].clone()
        if self.world > 1:
            t = t if self.dist.get_backend(self.group) != "gloo" else t.cpu()
            self.dist.all_reduce(t, group=self.group)
        return t.cpu().numpy()

    def mse(self) -> np.ndarray:
        t = self.mse_dev.clone()
        if self.world > 1:
            t = t if self.dist.get_backend(self.group) != "gloo" else t.cpu()
            self.dist.all_reduce(t, group=self.group)
        return t.cpu().numpy()

    def recon(self, out=None):
        """This rank's own rows of the current reconstruction (with `out`, a cubeio.CubeWriter: written there)."""
        own = self.recon_h[self.h_old][self.ext_lo:self.ext_lo + (self.g1 - self.g0)].numpy()
        return _write_or_copy(own, out)


__all__ = ["WavefrontRunner"]
