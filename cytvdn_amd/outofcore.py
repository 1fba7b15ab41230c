"""Out-of-core engine: a cube whose state does not fit in HBM, staged through pinned host memory.

The state (orig, recon, the compact accumulator arrays) lives in pinned host RAM.  Axis 0 is cut
into blocks; a block is copied to the GPU together with `k` halo rows per interior side, advanced by
`k` iterations there (temporal blocking: each iteration the sweep shrinks by one row per interior
side, because that row's neighbour is no longer current), and its own rows are copied back.  Two
staging buffers on two HIP streams overlap the PCIe copies of one block with the sweeps of the
other.  PCIe traffic per voxel is (10·(1+2k/R) arrays up + 9 arrays down) per k iterations instead of
per iteration, which is what makes the mode usable at all: PCIe Gen5 moves ≈ 55 GB/s per direction,
HBM ≈ 5 600 GB/s.

Results are bit-identical to the in-core engine: the sweeps are the same `tvdn_iterate_fused`
launches over sub-ranges of rows (SURVEY.md §7 hard part 6, §8d config 5).  Jia-Zhao BC only.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .engine import HipBackend, fista_ratios


class _BlockLayout:
    """Duck-typed layout of a staging buffer: `rows` x plane, every row 'own' (see HipBackend.set_block)."""

    def __init__(self, rows, plane, bc_mode):
        self.shape = (rows,) + tuple(plane)
        self.local_shape = self.shape
        self.row_lo, self.row_hi = 0, rows
        self.lo_mode, self.hi_mode = _lib.EDGE_BC, _lib.EDGE_BC
        self.bc_mode = bc_mode
        self.rank, self.world = 0, 1


def plan_blocks(n_rows: int, block_rows: int):
    """Own-row ranges [g0, g1) of the blocks."""
    block_rows = max(1, int(block_rows))
    return [(g, min(g + block_rows, n_rows)) for g in range(0, n_rows, block_rows)]


class HaloSwap:
    """Slab mode: for every host array (own rows [lo, hi), halo rows around them) my highest `depth` own rows go to the
    right neighbour's low halo rows and my lowest `depth` own rows to the left neighbour's high halo rows.

    Two uniform shifts, so every rank takes part in each at the same time (no chain of dependent messages):
      phase 1  everybody sends UP and receives its LOW halo rows   -- a pass needs them at once: `start` waits;
      phase 2  everybody sends DOWN and receives its HIGH halo rows -- a pass that streams upward needs them only for
               its last chunks: `start` posts the messages, `finish` waits, and the transfer hides under the pass.
    gloo moves host memory directly; RCCL stages the rows through HBM."""

    def __init__(self, dist, group, rank, world, device):
        self.dist, self.group, self.rank, self.world, self.device = dist, group, rank, world, device
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        self.via_dev = dist.get_backend(group) != "gloo"
        self.peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
        self.cuda = torch.device("cuda", device)
        self._pending = None

    def _snd(self, view):
        return view.to(self.cuda, non_blocking=False) if self.via_dev else view.contiguous()

    def _rcv(self, view, post):
        if not self.via_dev:
            return view
        buf = torch.empty(view.shape, dtype=view.dtype, device=self.cuda)
        post.append((view, buf))
        return buf

    def start(self, arrays, lo, hi, ext_lo, ext_hi, depth, overlap=True):
        self.finish()
        dist, dl, dh = self.dist, min(depth, ext_lo), min(depth, ext_hi)
        up, post_up, down, post_down = [], [], [], []
        for i, t in enumerate(arrays):
            if self.right is not None:
                up.append(dist.P2POp(dist.isend, self._snd(t[hi - depth:hi]), self.peer(self.right), self.group, tag=4 * i + 2))
            if self.left is not None:
                up.append(dist.P2POp(dist.irecv, self._rcv(t[lo - dl:lo], post_up), self.peer(self.left), self.group, tag=4 * i + 2))
        for i, t in enumerate(arrays):
            if self.left is not None:
                # sent while the pass runs, and the pass may write its new state into these very rows (in-place host
                # state of the wavefront engine): send a copy
                low = t[lo:lo + depth]
                down.append(dist.P2POp(dist.isend, self._snd(low) if self.via_dev else low.clone(), self.peer(self.left),
                                       self.group, tag=4 * i + 1))
            if self.right is not None:
                down.append(dist.P2POp(dist.irecv, self._rcv(t[hi:hi + dh], post_down), self.peer(self.right), self.group, tag=4 * i + 1))
        if up:
            for w in dist.batch_isend_irecv(up):
                w.wait()
        for view, buf in post_up:
            view.copy_(buf)
        if down:
            self._pending = (dist.batch_isend_irecv(down), post_down)
        if not overlap:
            self.finish()

    def finish(self):
        """Join phase 2 (no-op when nothing is pending)."""
        if self._pending is None:
            return
        works, post = self._pending
        self._pending = None
        for w in works:
            w.wait()
        for view, buf in post:
            view.copy_(buf)


class Row0Relay:
    """Staged slabs, Jia-Zhao, a cube whose FIRST row is not finite: the last rank closes the wrap of the reconstruction
    update with the accumulator upstream forms from the CURRENT recon of global row 0 (anisotropic.pyx:65-73; tvdn.h
    TVDN_EDGE_WRAP) -- at every iteration level of a temporally blocked pass.  Rank 0 computes row 0 of all levels in
    the first chunks of its pass; the last rank needs them in the last chunks of its own: one message per pass, posted
    early on both sides, so it hides under the pass.  gloo moves host memory, RCCL device memory."""

    def __init__(self, dist, group, rank, world, planes):
        self.dist, self.group, self.rank, self.world, self.planes = dist, group, rank, world, planes
        self.via_dev = dist.get_backend(group) != "gloo"
        self.peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
        self._work = self._buf = None
        self.n = 0

    def _staging(self, n):
        p = self.planes[0]
        return torch.empty((n,) + tuple(p.shape), dtype=p.dtype, device=p.device if self.via_dev else "cpu")

    def post_recv(self, n):
        """Last rank, at the start of a pass of n - 1 levels."""
        self.n, self._buf = n, self._staging(n)
        self._work = self.dist.irecv(self._buf, self.peer(0), group=self.group, tag=1000)

    def send(self, n):
        """Rank 0, once planes[0 .. n) hold row 0 of levels 0 .. n - 1 (the stream that wrote them is joined here)."""
        torch.cuda.current_stream(self.planes[0].device).synchronize()
        self._buf = self._staging(n)
        for j in range(n):
            self._buf[j].copy_(self.planes[j])
        torch.cuda.current_stream(self.planes[0].device).synchronize()
        self._work = self.dist.isend(self._buf, self.peer(self.world - 1), group=self.group, tag=1000)

    def wait_recv(self):
        """Last rank, before its first sweep at the cube's top face."""
        if self._work is None:
            return
        self._work.wait()
        for j in range(self.n):
            self.planes[j].copy_(self._buf[j])
        self._work = self._buf = None

    def finish(self):
        """Rank 0, at the end of the pass."""
        if self._work is not None:
            self._work.wait()
            self._work = self._buf = None


def exchange_halo_rows(dist, group, rank, world, device, arrays, lo, hi, ext_lo, ext_hi, depth):
    """Blocking form of HaloSwap: both shifts completed on return."""
    HaloSwap(dist, group, rank, world, device).start(arrays, lo, hi, ext_lo, ext_hi, depth, overlap=False)


class StagedRunner:
    """Runs the denoise loop on a host-resident cube (or, with `world` > 1, on this rank's slab of it)
    through a few staging buffers on one GPU.

    Slab mode: `datacube` holds this rank's own rows [row0, row0+rows) of a cube with `global_rows` rows;
    the host arrays carry up to `k` extra rows per interior side, refreshed from the neighbouring ranks
    before every pass (k rows of recon and of every accumulator-state array: a temporally blocked pass needs
    the neighbours' full state k rows deep, against one row of recon per iteration for the in-core engine)."""

    def __init__(self, datacube: np.ndarray, fista: bool, clip, lam_mu, bc_mode: int = 2, device: int = 0,
                 block_rows: int = 32, k: int = 8, max_iters: int = 1, reference: np.ndarray = None,
                 pin: bool = True, global_rows: int = None, row0: int = 0, group=None, world: int = 1, rank: int = 0,
                 n_stages: int = 3, exact_wrap: bool = False):
        if bc_mode != 2:
            raise NotImplementedError("the staged engine supports the Jia-Zhao boundary condition (BC_mode=2) only")
        own_shape = tuple(int(s) for s in datacube.shape)
        self.nd = len(own_shape)
        self.dtype = datacube.dtype
        self.fista = bool(fista)
        self.device = int(device)
        self.k = max(1, int(k))
        self.N0 = int(own_shape[0] if global_rows is None else global_rows)
        self.row0, self.own = int(row0), own_shape[0]
        self.world, self.rank, self.group = int(world), int(rank), group
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            if self.own < self.k:
                raise ValueError(f"a slab needs at least k = {self.k} rows (it has {self.own})")
        self.ext_lo = min(self.k, self.row0) if self.world > 1 else 0
        self.ext_hi = min(self.k, self.N0 - (self.row0 + self.own)) if self.world > 1 else 0
        self.base = self.row0 - self.ext_lo                 # global index of host row 0
        self.shape = (self.ext_lo + self.own + self.ext_hi,) + own_shape[1:]
        self.blocks = [(self.row0 + a, self.row0 + b) for a, b in plan_blocks(self.own, block_rows)]
        self.max_iters = max(1, int(max_iters))
        tdt = torch.float32 if self.dtype == np.float32 else torch.float64
        plane = self.shape[1:]
        own_sl = slice(self.ext_lo, self.ext_lo + self.own)

        def host(fill=None):
            t = torch.empty(self.shape, dtype=tdt, pin_memory=pin)
            t.zero_()
            if fill is not None:
                t[own_sl].copy_(torch.from_numpy(fill))
            return t

        as_source = np.ascontiguousarray

        # host state: orig, recon old/new, per axis up to two state arrays old/new
        if pin:     # page-locked memory cannot swap: refuse here, whoever the caller is, what this host cannot hold
            from .planner import check_host_fits
            n_host = 3 + 2 * self.nd * (2 if self.fista else 1) + (reference is not None)
            check_host_fits(dict(mode="trapezoid", k=self.k,
                                 host_bytes_per_rank=n_host * int(np.prod(self.shape)) * self.dtype.itemsize))
        self.orig_h = host(as_source(datacube))
        self.recon_h = [host(as_source(datacube)), host()]
        n_state = 2 if self.fista else 1
        self.state_h = [[[host() for _ in range(n_state)] for _ in range(self.nd)] for _ in range(2)]  # [old/new][axis][j]
        self.ref_h = host(np.ascontiguousarray(reference)) if reference is not None else None
        self.h_old = 0
        self.d_form = self.fista
        self.tk_prev = 0.0
        self.iters_done = 0
        self._carry_buf = None
        if self.world > 1:
            self._exchange([self.orig_h], self.k)           # the input's halo rows never change

        max_rows = min(self.shape[0], max(g1 - g0 for g0, g1 in self.blocks) + 2 * self.k)
        self.stages = []
        for _ in range(max(2, int(n_stages))):
            be = HipBackend(_BlockLayout(max_rows, plane, bc_mode), self.dtype, self.fista, device=self.device,
                            max_iters=self.max_iters + 1, private_ctx=True)       # last sums row: discard slot
            be.set_params(clip, lam_mu)
            be.stream = torch.cuda.Stream(device=self.device)
            be.ref = torch.empty_like(be.orig) if reference is not None else None
            be.mse = torch.zeros(self.max_iters + 1, dtype=torch.float64, device=be.orig.device) if reference is not None else None
            be.tmp = torch.zeros(1, dtype=torch.float64, device=be.orig.device)
            self.stages.append(be)
        # `exact_wrap` (single process): the block at the cube's top face forms the wrapped axis-0 accumulator from the
        # recon of global row 0 at each level (TVDN_EDGE_WRAP), which the block at the bottom face sets aside as it
        # goes; see wavefront.py and engine.py ("Non-finite data")
        self.row0 = None
        if exact_wrap and self.world == 1:
            self.row0 = [torch.empty(tuple(plane), dtype=tdt, device=self.stages[0].orig.device) for _ in range(self.k + 1)]
        self.bytes_h2d = 0
        self.bytes_d2h = 0
        if self.ref_h is not None:  # MSE[0]: input against reference (cyTVDN.py:124-125), block by block
            be = self.stages[0]
            with torch.cuda.stream(be.stream):
                for g0, g1 in self.blocks:
                    n = g1 - g0
                    be.orig[:n].copy_(self.orig_h[g0 - self.base:g1 - self.base], non_blocking=True)
                    be.ref[:n].copy_(self.ref_h[g0 - self.base:g1 - self.base], non_blocking=True)
                    self._sse(be, be.orig[:n], be.ref[:n], 0)
            be.stream.synchronize()

    def _exchange(self, arrays, depth):
        exchange_halo_rows(self.dist, self.group, self.rank, self.world, self.device, arrays, self.ext_lo,
                           self.ext_lo + self.own, self.ext_lo, self.ext_hi, depth)

    def _sse(self, be, a, b, slot):
        _lib.check(_lib.lib().tvdn_sum_square_error(be.ctx, be.code, self.nd, _lib.shape_arr(a.shape), a.data_ptr(),
                                                    b.data_ptr(), be.tmp.data_ptr(), _lib.current_stream(self.device)))
        be.mse[slot:slot + 1] += be.tmp

    # one super-step: `ratios` holds the tk ratio of each iteration (None = unaccelerated)
    def _superstep(self, ratios, slot0):
        kk = len(ratios)
        N0, base = self.N0, self.base
        old, new = self.h_old, self.h_old ^ 1
        if self.world > 1:  # neighbours' rows of the current state, kk deep
            arrays = [self.recon_h[old]] + [t for q in range(self.nd) for t in self.state_h[old][q][: (2 if self.d_form else 1)]]
            self._exchange(arrays, kk)
        discard = self.max_iters                      # sums row nobody reads
        form_after = tk_after = None
        prev = None                                   # (s0, s1, carry-ready event, first carried row) of the previous block
        row0_ready = None                             # exact_wrap: the bottom block has set row 0 of every level aside
        for bi, (g0, g1) in enumerate(self.blocks):
            be = self.stages[bi % len(self.stages)]
            s0, s1 = max(0, g0 - kk), min(N0, g1 + kk)
            rows = s1 - s0
            lo_edge, hi_edge = (s0 == 0), (s1 == N0)
            whole = lo_edge and hi_edge
            stash = self.row0 is not None and not whole
            with torch.cuda.stream(be.stream):
                be.set_block(rows, _lib.EDGE_BC if (whole or not hi_edge) else (_lib.EDGE_WRAP if stash else _lib.EDGE_ZERO))
                be.set_form(self.d_form, self.tk_prev)
                # ---- bring in the block with its halo rows ------------------------------------------------
                # Rows it shares with the previous block (that block's top 2k rows, still at the state of the
                # pass start there) are forwarded device-to-device through a small carry buffer instead of
                # crossing PCIe twice; only the rest is uploaded.
                h0, h1 = s0 - base, s1 - base            # the staged rows inside the host arrays
                ov = 0                                   # leading rows that come from the carry buffer
                if prev is not None and prev[1] > s0:
                    ov = min(prev[1], s1) - s0
                dst = [be.orig, be.recon[be.cur]] + [t for arrs in be.state_tensors() for t in arrs]
                src_h = [self.orig_h, self.recon_h[old]] + [self.state_h[old][q][j] for q, arrs in
                                                            enumerate(be.state_tensors()) for j in range(len(arrs))]
                for t, hsrc in zip(dst, src_h):
                    t[ov:rows].copy_(hsrc[h0 + ov:h1], non_blocking=True)
                self.bytes_h2d += len(dst) * (rows - ov) * self._row_bytes()
                if ov:
                    be.stream.wait_event(prev[2])
                    c_off = s0 - prev[3]                 # where row s0 sits inside the carry buffer
                    for t, c in zip(dst, self._carry(len(dst), be)):
                        t[:ov].copy_(c[c_off:c_off + ov], non_blocking=True)
                if self.ref_h is not None:
                    be.ref[:g1 - g0].copy_(self.ref_h[g0 - base:g1 - base], non_blocking=True)
                    self.bytes_h2d += (g1 - g0) * self._row_bytes()
                if bi + 1 < len(self.blocks):            # my top rows, untouched yet, for the next block
                    c0 = max(s0, s1 - 2 * kk)
                    for t, c in zip(dst, self._carry(len(dst), be)):
                        c[:s1 - c0].copy_(t[c0 - s0:rows], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(be.stream)
                    prev = (s0, s1, ev, c0)
                # ---- k iterations on a shrinking range of rows ----------------------------------------
                v0, v1 = 0, rows                      # rows whose state is current
                o0, o1 = g0 - s0, g1 - s0             # own rows, local
                if stash and lo_edge:
                    self.row0[0].copy_(be.recon_tensor()[0])
                if stash and hi_edge:
                    be.stream.wait_event(row0_ready)  # recorded by the bottom block, which is always issued first
                for j, tk in enumerate(ratios):
                    a = v0 if lo_edge else v0 + 1
                    b = v1 if hi_edge else v1 - 1
                    slot = slot0 + j
                    be._args.wrap_recon = self.row0[j].data_ptr() if (stash and hi_edge) else None
                    # the sums count own rows only: halo rows go to a discard slot
                    if a < o0:
                        be.step(tk, discard, rows=(a, o0), accumulate=True)
                    be.step(tk, slot, rows=(o0, o1), accumulate=True)
                    if o1 < b:
                        be.step(tk, discard, rows=(o1, b), accumulate=True)
                    be.flip()
                    v0, v1 = a, b
                    if stash and lo_edge:
                        self.row0[j + 1].copy_(be.recon_tensor()[0])
                    if self.ref_h is not None:
                        self._sse(be, be.ref[:o1 - o0], be.recon_tensor()[o0:o1], slot + 1)
                if stash and lo_edge:
                    row0_ready = torch.cuda.Event()
                    row0_ready.record(be.stream)
                be._args.wrap_recon = None
                # ---- download the own rows ----------------------------------------------------------------
                self.recon_h[new][g0 - base:g1 - base].copy_(be.recon_tensor()[o0:o1], non_blocking=True)
                n_down = 1
                st = be.state_tensors()
                for q, arrs in enumerate(st):
                    for j, t in enumerate(arrs):
                        self.state_h[new][q][j][g0 - base:g1 - base].copy_(t[o0:o1], non_blocking=True)
                        n_down += 1
                self.bytes_d2h += n_down * (g1 - g0) * self._row_bytes()
                form_after, tk_after = be.d_form, be.tk_prev
        for be in self.stages:
            be.stream.synchronize()
        self.h_old = new
        self.d_form, self.tk_prev = form_after, tk_after
        self.iters_done += kk

    def _carry(self, n_arrays, be):
        """Device buffer of 2k rows per staged array, handed from one block to the next."""
        if self._carry_buf is None or len(self._carry_buf) < n_arrays:
            rows = min(2 * self.k, be.orig.shape[0])
            self._carry_buf = [torch.empty((rows,) + tuple(be.orig.shape[1:]), dtype=be.orig.dtype, device=be.orig.device)
                               for _ in range(1 + 1 + 2 * self.nd)]
        return self._carry_buf[:n_arrays]

    def _row_bytes(self):
        return int(np.prod(self.shape[1:])) * self.dtype.itemsize

    def run(self, n_fista: int, n_plain: int, on_superstep=None):
        """n_fista FISTA iterations then n_plain unaccelerated ones, k at a time.
        `on_superstep(first_slot, count)` may return True to stop the current phase."""
        slot = self.iters_done
        ratios = [float(r) for r in fista_ratios(n_fista)]
        i = 0
        while i < n_fista:
            grp = ratios[i:i + self.k]
            self._superstep(grp, slot + i)
            i += len(grp)
            if on_superstep is not None and on_superstep(slot + i - len(grp), len(grp)):
                break
        slot += n_fista
        j = 0
        while j < n_plain:
            n = min(self.k, n_plain - j)
            self._superstep([None] * n, slot + j)
            j += n
            if on_superstep is not None and on_superstep(slot + j - n, n):
                break

    def _allreduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            t = t if self.dist.get_backend(self.group) != "gloo" else t.cpu()
            self.dist.all_reduce(t, group=self.group)
        return t

    def sums(self) -> np.ndarray:
        """[max_iters, 3] f64: b_norm, sum|delta|, sum|old| per iteration, summed over the blocks (and, in
        slab mode, over the ranks)."""
        t = sum(be.sums[: self.max_iters] for be in self.stages)
        return self._allreduce(t.clone()).cpu().numpy()

    def mse(self) -> np.ndarray:
        return self._allreduce(sum(be.mse for be in self.stages).clone()).cpu().numpy()

    def recon(self):
        """This rank's own rows of the current reconstruction."""
        return self.recon_h[self.h_old][self.ext_lo:self.ext_lo + self.own].numpy().copy()


__all__ = ["StagedRunner", "plan_blocks"]
