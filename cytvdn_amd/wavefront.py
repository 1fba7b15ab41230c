"""Out-of-core engine, wavefront schedule: k iterations per PCIe round trip with NO redundant sweeps.

`outofcore.StagedRunner` advances a block k iterations by letting the swept range shrink one row per
iteration on each interior side (a trapezoid in row x iteration space): simple, but at k ~ block height half
of the sweeps are spent on halo rows.  Here the blocks are parallelograms instead: the cube streams
through the GPU once per pass, in chunks of R rows, and iteration level j+1 trails level j by one row --
chunk c computes, for j = 0..k-1, rows [cR-(j+1), (c+1)R-(j+1)) of level j+1 from level j -- so every row of
every level is computed exactly once and crosses PCIe once per k iterations (10 arrays up, 9 down).

Each level keeps a ring of R+2 rows per state array in HBM (recon_j and one accumulator array per axis; in the
compact FISTA state level j's `d_j` also serves as `d_prev` of level j+1's update), plus one ring of the input.
The sweeps are the same `tvdn_iterate_fused` launches as everywhere else, told that row g of each array lives at
slot g % ring_rows (tvdn.h): rows are never moved inside HBM, and the arithmetic -- and the bits -- are those of the
in-core engine.  Jia-Zhao and (single process) periodic BC; slabs across
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


IO_STREAMS = 1      # HIP streams per PCIe direction of the wavefront engine (TVDN_IO_STREAMS overrides)
WINDOW_SLACK = 2    # rows a level window holds beyond the chunk height (planner.wavefront_windows uses the same)


class _Ring:
    """Ring of `cap` row-planes of one array: global row g lives at slot g % cap (tvdn.h, ring_rows).  A level keeps
    the R+2 rows the next level's launch reads; nothing is ever moved (round 2's first version slid the windows
    down by copying: 2/R of a sweep's traffic on top of it, 40 % at 2-row chunks)."""

    def __init__(self, cap, plane, tdt, dev):
        self.buf = torch.zeros((cap,) + tuple(plane), dtype=tdt, device=dev)
        self.cap = int(cap)

    def row(self, g):
        return self.buf[g % self.cap]

    def rows(self, g0, g1):
        return [self.buf[g % self.cap] for g in range(g0, g1)]

    def ptr(self):
        return self.buf.data_ptr()


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
        self.R = max(1, int(chunk_rows))
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
            if fill is not None:
                t[own_sl].copy_(torch.from_numpy(fill))
            return t

        as_source = np.ascontiguousarray

        # Host state.  A pass reads row r of the old state (upload) at least k rows ahead of where it writes the new
        # state (download), so old and new can be the SAME pinned arrays: 1 + 1 + nd x n_state of them (10 for 4-D
        # FISTA) instead of 19 -- 2.5 TiB instead of 4.9 TiB for BASELINE config 5, which is what fits a 3 TB host.
        # `host_inplace=False` keeps separate old/new arrays (cross-check).
        n_state = 2 if self.fista else 1
        if pin:     # page-locked memory cannot swap: refuse here, whoever the caller is, what this host cannot hold
            from .planner import check_host_fits
            n_host = (2 + self.nd * n_state) * (1 if host_inplace else 2) - (0 if host_inplace else 1) + (reference is not None)
            check_host_fits(dict(mode="wavefront", k=self.k,
                                 host_bytes_per_rank=n_host * int(np.prod(self.shape)) * self.dtype.itemsize))
        self.orig_h = host(as_source(datacube))
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
        self.cap = cap
        self.Rw = [_Ring(cap, plane, tdt, dev) for _ in range(K + 1)]
        self.Aw = [[_Ring(cap, plane, tdt, dev) for _ in range(self.nd)] for _ in range(K + 2)]  # index level + 1
        self.Ow = _Ring(self.R + K + 3, plane, tdt, dev)
        self.Fw = _Ring(self.R + K + 3, plane, tdt, dev) if reference is not None else None   # reference_data rows
        n_in = 3 + 2 * self.nd
        n_out = 1 + 2 * self.nd
        self.inbox = [[torch.empty((self.R,) + tuple(plane), dtype=tdt, device=dev) for _ in range(n_in)] for _ in range(2)]
        self.outbox = [[torch.empty((self.R,) + tuple(plane), dtype=tdt, device=dev) for _ in range(n_out)] for _ in range(2)]
        # HIP streams per PCIe direction.  One each is best: up and down together already hold the link at 32 + 29 GB/s
        # (256 MiB planes, 32x1024x256x256, 2 rows x k = 24: 19.1 Gvoxel-iters/s with 1 stream per direction, 18.4
        # with 2, 13.9 with 4); the knob stays for other hosts.
        # Downloads: the runtime's DMA copies by default.  TVDN_WF_DOWN=kernel sends a chunk's rows home with ONE capped
        # copy launch writing pinned memory instead (tvdn_copy_many, TVDN_WF_IO_BLOCKS workgroups): it never falls into
        # the 4x slower mode whole bursts of DMA downloads show on some boxes (profiles/r02_wavefront_rings.txt) but
        # slows the sweeps beside it by a third -- better at 2-4 row chunks, worse at 16; uploads by kernel are always worse.
        self.down_kernel = os.environ.get("TVDN_WF_DOWN", "dma") == "kernel"
        self.io_blocks = int(os.environ.get("TVDN_WF_IO_BLOCKS", "16"))
        n_io = max(1, int(os.environ.get("TVDN_IO_STREAMS", str(IO_STREAMS))))
        # Upload and download streams come from DIFFERENT priority pools: the runtime multiplexes streams onto a few
        # hardware queues per priority level, and two streams that land on the same one execute in submission order
        # (csrc/tvdn_common.hpp make_stream; tvdn_run's pipelined download waited behind every queued sweep that way).
        # The sweeps run on the default stream, which has a queue of its own.  TVDN_WF_DOWN_PRIO=0: one pool (measurement).
        down_prio = int(os.environ.get("TVDN_WF_DOWN_PRIO", "-1"))
        self.ups = [torch.cuda.Stream(device=dev) for _ in range(n_io)]
        self.downs = [torch.cuda.Stream(device=dev, priority=down_prio) for _ in range(n_io)]
        self._args = _lib.IterArgs()
        # Jia-Zhao, `exact_wrap` (single process): the sweeps at the cube's top face form the wrapped axis-0 accumulator
        # from the recon of global row 0 AT THEIR OWN LEVEL (TVDN_EDGE_WRAP) instead of taking it as zero, which it is
        # only while row 0 is finite (engine.py, "Non-finite data").  Row 0 of every level is computed at the start of a
        # pass and has long left its window when the top is reached, so one plane per level is kept aside.
        # Across ranks (staged slabs) rank 0 stashes the planes and sends them to the last rank once per pass (Row0Relay).
        self.row0 = self._relay = None
        if exact_wrap and not self.periodic and (self.world == 1 or self.rank in (0, self.world - 1)):
            self.row0 = [torch.empty(tuple(plane), dtype=tdt, device=dev) for _ in range(K + 1)]
            if self.world > 1:
                from .outofcore import Row0Relay
                self._relay = Row0Relay(self.dist, self.group, self.rank, self.world, self.row0)

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
        """Rows are global row numbers of a virtual N0-row array of which each level's ring holds the R+2 current ones."""
        N0 = self.N0
        A = self._args
        A.dtype, A.ndim = self.code, self.nd
        A.shape[0] = N0
        for i, s in enumerate(self.shape[1:]):
            A.shape[i + 1] = s
        A.row_lo, A.row_hi = 0, N0
        A.sweep_lo, A.sweep_hi = a, b
        A.ring_rows, A.orig_ring_rows = self.cap, self.Ow.cap
        A.lo_mode = _lib.EDGE_BC
        # the top face is reached only where it is the cube's own (Jia-Zhao: wrapped accumulator zero, or exact from the
        # stashed row 0); periodic runs extend the cube instead and never sweep a row next to row_hi
        A.hi_mode = _lib.EDGE_BC if self.periodic else _lib.EDGE_ZERO
        A.wrap_recon = None
        if self.row0 is not None and self.g1 == N0:        # the rank (or the one process) that owns the cube's top face
            A.hi_mode, A.wrap_recon = _lib.EDGE_WRAP, self.row0[j].data_ptr()
        A.bc_mode = self.bc
        A.mode = mode
        A.tk, A.tk_prev = float(tk or 0.0), float(tk_prev)
        A.accumulate = 1
        A.orig = self.Ow.ptr()
        A.recon_in = self.Rw[j].ptr()
        A.recon_out = self.Rw[j + 1].ptr()
        for q in range(self.nd):
            cur, prv, nxt = self.Aw[j + 1][q].ptr(), self.Aw[j][q].ptr(), self.Aw[j + 2][q].ptr()
            A.b_in[q] = A.b_out[q] = A.d_in[q] = A.d_out[q] = A.dprev_in[q] = None
            A.clip[q], A.lambda_mu[q] = self.clip[q], self.lam_mu[q]
            if mode == _lib.ITER_FISTA_D:
                A.d_in[q], A.dprev_in[q], A.d_out[q] = cur, prv, nxt
            elif mode == _lib.ITER_FISTA_D_TO_PLAIN:
                A.d_in[q], A.dprev_in[q], A.b_out[q] = cur, prv, nxt
            else:
                A.b_in[q], A.b_out[q] = cur, nxt
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
            modes.append(_lib.iter_mode(tk is not None, forms[j]))   # the library's rule (ValueError: FISTA after plain)
            forms.append(tk is not None)
            tkp.append(prev_ratio)
            if tk is not None:
                prev_ratio = tk
        n_in_state = 2 if forms[0] else 1
        n_out_state = 2 if forms[kk] else 1
        n_chunks = (E1 - E0 + kk + R - 1) // R

        def lo_bound(level):    # lowest row that can be brought to `level` (an artificial face loses a row per level)
            return E0 + level if art_lo else 0

        def hi_bound(level):
            return E1 - level if art_hi else N0
        in_ready, in_free = [None, None], [None, None]
        out_ready, out_free = [None, None], [None, None]
        trace = [] if os.environ.get("TVDN_WF_TRACE") else None   # (kind, chunk, start event, end event) per phase

        def mark(stream):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
            return ev

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
                    t0 = mark(st) if trace is not None else None
                    for dst, src in pairs[si::len(self.ups)]:
                        dst[:n].copy_(src[u0 - hb:u1 - hb], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(st)
                    evs.append(ev)
                    if trace is not None:
                        trace.append(("up", c, t0, mark(st)))
            self.bytes_h2d += len(pairs) * n * self.row_bytes
            in_ready[c % 2] = evs

        relay = self._relay
        stash = self.row0 is not None and g0 == 0           # this rank computes global row 0
        stashed, sent = 0, False
        if relay is not None and self.rank == self.world - 1:
            relay.post_recv(kk + 1)
        upload(0)
        for c in range(n_chunks):
            upload(c + 1)                                   # next chunk crosses PCIe while this one is swept
            if relay is not None and self.rank == self.world - 1 and E0 + (c + 1) * R - 1 >= N0:
                relay.wait_recv()                           # this chunk's level 0 reaches the top face (no-op afterwards)
            u0, u1 = E0 + c * R, min(E0 + (c + 1) * R, E1)
            if u0 < u1:
                n = u1 - u0
                box = self.inbox[c % 2]
                for ev in in_ready[c % 2]:
                    main.wait_event(ev)
                rings = [self.Ow, self.Rw[0]]
                for q in range(nd):
                    rings.append(self.Aw[1][q])                                    # level 0: d_k (or b)
                    if n_in_state == 2:
                        rings.append(self.Aw[0][q])                                # level -1: d_k-1
                srcs = list(box[:len(rings)])
                if self.Fw is not None:
                    rings.append(self.Fw)
                    srcs.append(box[-1])
                _lib.copy_many([(rg.row(g), src[g - u0]) for rg, src in zip(rings, srcs) for g in range(u0, u1)], self.device)
                if stash and u0 == 0:
                    self.row0[0].copy_(self.Rw[0].row(0))
                if self.Fw is not None and self.iters_done == 0:
                    # MSE[0]: the input against the reference (cyTVDN.py:124-125), own rows
                    for g in range(max(u0, g0), min(u1, g1)):
                        self._sse(self.Rw[0].row(g)[None], self.Fw.row(g)[None], 0)
                ev = torch.cuda.Event()
                ev.record(main)
                in_free[c % 2] = ev
            # the wavefront: level j+1 trails level j by one row
            t0 = mark(main) if trace is not None else None
            for j in range(kk):
                a = max(lo_bound(j + 1), E0 + c * R - (j + 1))
                b = min(hi_bound(j + 1), E0 + (c + 1) * R - (j + 1))
                if a >= b:
                    continue
                # the sums count own rows only: rows of the neighbours' halo go to a discard slot
                for x0, x1, slot in ((a, min(b, g0), discard), (max(a, g0), min(b, g1), slot0 + j), (max(a, g1), b, discard)):
                    if x0 < x1:
                        self._launch(j, x0, x1, ratios[j], tkp[j], modes[j], slot)
                        if stash and x0 == 0:
                            self.row0[j + 1].copy_(self.Rw[j + 1].row(0))
                            stashed += 1
                        if self.Fw is not None and slot != discard:
                            for g in range(x0, x1):
                                self._sse(self.Fw.row(g)[None], self.Rw[j + 1].row(g)[None], slot + 1)
            if relay is not None and self.rank == 0 and stashed == kk and not sent:
                relay.send(kk + 1)                          # row 0 of every level of this pass is final: off to the last rank
                sent = True
            if trace is not None:
                trace.append(("sweep", c, t0, mark(main)))
            # own rows that have reached the last level go home
            a = max(g0, E0 + c * R - kk)
            b = min(g1, E0 + (c + 1) * R - kk)
            if a < b:
                n = b - a
                box = self.outbox[c % 2]
                for ev in out_free[c % 2] or ():
                    main.wait_event(ev)
                rings = [self.Rw[kk]]
                for q in range(nd):
                    rings.append(self.Aw[kk + 1][q])
                    if n_out_state == 2:
                        rings.append(self.Aw[kk][q])
                _lib.copy_many([(box[i][g - a], rg.row(g)) for i, rg in enumerate(rings) for g in range(a, b)], self.device)
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
                        t0 = mark(st) if trace is not None else None
                        mine = [(dst[a - hb:b - hb], src[:n]) for dst, src in pairs[si::len(self.downs)]]
                        if self.down_kernel:
                            _lib.copy_many(mine, self.device, self.io_blocks)   # posted writes into pinned memory, one launch
                        else:
                            for dst, src in mine:
                                dst.copy_(src, non_blocking=True)
                        ev2 = torch.cuda.Event()
                        ev2.record(st)
                        evs.append(ev2)
                        if trace is not None:
                            trace.append(("down", c, t0, mark(st)))
                self.bytes_d2h += len(pairs) * n * self.row_bytes
                out_free[c % 2] = evs
        for st in self.downs:
            st.synchronize()
        main.synchronize()
        if relay is not None:
            relay.finish()
        if trace:
            # timeline of the pass relative to its first event (ms): start-end of every phase, to see what overlaps what
            ref = trace[0][2]
            import sys
            for kind, c, e0, e1 in sorted(trace, key=lambda t: ref.elapsed_time(t[2])):
                print(f"wf-trace {kind:5s} chunk {c:4d}  {ref.elapsed_time(e0):9.2f} -> {ref.elapsed_time(e1):9.2f} ms", file=sys.stderr)
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

    def recon(self):
        """This rank's own rows of the current reconstruction."""
        return self.recon_h[self.h_old][self.ext_lo:self.ext_lo + (self.g1 - self.g0)].numpy().copy()


__all__ = ["WavefrontRunner"]
