"""Multi-GPU denoising with the reference's argument vocabulary: one process per GPU, one slab each.

This is what replaces the reference's `cyTVMPI` loop (cyTVDN/mpi.py:314-434) -- and adds what that
loop lacks (README.md:34 to-do): FISTA, 3-D data, globally reduced b_norm / delta_recon traces and a
global stopping criterion.  Launch with torchrun (RCCL over xGMI); every rank calls

    own, b_norm, delta_recon = denoise_slabs(my_rows, global_shape, mu, iterations, FISTA=True)

`my_rows` are this rank's rows of axis 0 (SlabLayout(global_shape, rank, world).g0 .. g1).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .engine import HipBackend, SlabLayout, SlabRunner


def slab_rows(global_shape, rank: int, world: int, bc_mode: int = 2):
    """(g0, g1): the rows of axis 0 owned by `rank`."""
    lay = SlabLayout(tuple(global_shape), rank, world, bc_mode)
    return lay.g0, lay.g1


def denoise_slabs(my_rows, global_shape, mu, iterations=10, FISTA=True, stopping_relative_change=None,
                  BC_mode=2, lam=None, group=None, device=None, backend_factory=None, staged=None):
    """Slab-parallel denoise3D/denoise4D.  Returns (own rows of recon, b_norm, delta_recon); the traces
    are global (all-reduced) and identical on every rank.  Semantics of `iterations` ([nF, nU] hybrid),
    `lam` defaults and the stopping rule follow cyTVDN/cyTVDN.py:67-68 / :294-295, :99-108, :189-195.

    `staged=(rows, k)` keeps each rank's slab in pinned host memory and streams it through the GPU, k iterations
    per PCIe round trip -- the library's streamed loop (tvdn_run with a tvdn_slab_io, csrc/tvdn_stream.hip), which
    calls back here for the k rows of state it swaps with its neighbours per pass, the sums and the wrap row; with a
    stopping rule one iteration per pass.  Interior rows of the slab stay resident in HBM as far as they fit
    (`staged=(rows, k, n)` caps them at n): they never cross PCIe and take no page-locked host memory.  `staged="auto"`
    leaves the choice to `planner.plan_run` (rank 0 plans for all): resident slabs when they fit their GPUs, else the fastest
    streamed plan the host memory holds.  For cubes whose state exceeds the HBM of the GPUs at hand (BASELINE
    config 5).  Without it the slab must fit in HBM."""
    import torch.distributed as dist
    if not dist.is_initialized():
        raise RuntimeError("initialise torch.distributed first (backend 'nccl' = RCCL on ROCm)")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    nd = len(global_shape)
    if nd not in (3, 4):
        raise TypeError("No matching signature found")
    is_t = isinstance(my_rows, torch.Tensor)
    dtype = np.dtype({torch.float32: np.float32, torch.float64: np.float64}[my_rows.dtype]) if is_t else my_rows.dtype
    assert dtype in (np.float32, np.float64), "datacube must be floating point datatype."
    mu = np.asarray(mu)
    if lam is None:
        lam = mu * 1.0 / 32.0 if nd == 4 else mu / 16.0
    lam = np.asarray(lam)
    assert lam.dtype == dtype, "Lambda must have same dtype as datacube."
    if BC_mode == 1:
        raise NotImplementedError("BC_mode=1 (mirror) is undefined behaviour upstream (utils.pyx:117-120)")
    lay = SlabLayout(tuple(int(s) for s in global_shape), rank, world, int(BC_mode))
    if world > 1 and int(BC_mode) == 2 and staged is None and lay.own_rows == my_rows.shape[0]:
        # exact Jia-Zhao wrap for a cube whose FIRST row is not finite (see engine.py): every rank must agree
        bad = 0
        if rank == 0 and my_rows.shape[0] > 0:
            first = my_rows[0]
            bad = int(not bool(torch.isfinite(first).all() if is_t else np.isfinite(first).all()))
        flag = torch.tensor([bad], dtype=torch.int32)
        if dist.get_backend(group) == "nccl":
            flag = flag.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag.item()):
            lay = SlabLayout(lay.shape, rank, world, 2, wrap_row=True)
    if tuple(my_rows.shape) != (lay.own_rows,) + tuple(global_shape[1:]):
        raise ValueError(f"rank {rank} owns rows {lay.g0}..{lay.g1}: expected shape "
                         f"{(lay.own_rows,) + tuple(global_shape[1:])}, got {tuple(my_rows.shape)}")
    unacc = not FISTA
    if type(iterations) in (list, tuple):
        FISTA, unacc = True, True
        n_f, n_p = int(iterations[0]), int(iterations[1])
    else:
        n_f, n_p = int(iterations * FISTA), int(iterations * (not FISTA))
    n = n_f + n_p
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    if isinstance(staged, str) or staged is True:
        # staged="auto": the planner decides -- slabs resident in HBM when they fit, else every rank streams its slab with the
        # (rows, k, resident rows) that are fastest by the library's model AMONG those whose page-locked state fits the host
        # these ranks share (planner.plan_run).  Rank 0 plans, everybody follows: k must be the same on every rank.
        if staged is not True and staged != "auto":
            raise ValueError(f"staged must be (rows, k[, resident rows]), 'auto' or None, got {staged!r}")
        from .planner import plan_run
        plan = [None]
        if rank == 0:
            p = plan_run(tuple(int(v) for v in global_shape), dtype, FISTA, world, stop=stopping_relative_change is not None,
                         device=device, swap_through_host=dist.get_backend(group) != "nccl")
            plan[0] = None if p["mode"] in ("slabs", "in-core") else (p["mode"], p.get("chunk_rows"), p.get("k"),
                                                                       p.get("resident_rows_per_rank", -1), p["why"])
        dist.broadcast_object_list(plan, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
        if plan[0] is None:
            staged = None
        elif plan[0][0] not in ("slabs+wavefront", "wavefront"):
            raise MemoryError(f"no plan for this cube on {world} ranks: {plan[0][4]}")
        else:
            staged = (int(plan[0][1]), int(plan[0][2]), int(plan[0][3]))
    if staged is not None:
        exact_wrap = False
        if world > 1 and int(BC_mode) == 2:       # a non-finite first row: every rank must know (see engine.py)
            bad = 0
            if rank == 0 and my_rows.shape[0] > 0:
                first = my_rows[0]
                bad = int(not bool(torch.isfinite(first).all() if is_t else np.isfinite(first).all()))
            flag = torch.tensor([bad], dtype=torch.int32)
            if dist.get_backend(group) == "nccl":
                flag = flag.to(torch.device("cuda", device))
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            exact_wrap = bool(int(flag.item()))
        return _denoise_slabs_staged(my_rows, lay, dtype, mu, lam, FISTA, unacc, n_f, n_p, stopping_relative_change,
                                     group, device, staged, rank, world, exact_wrap)
    # The overlapped exchange (edge rows first, transfer on a side stream) is used only on a process group on which it has
    # reproduced a single-GPU run bit for bit in this process (selfcheck_exchange, cached per group); otherwise the blocking
    # exchange, loudly.  The same goes for WHAT the rows are sent out of and received into: a slab's state of 2 GiB or more
    # lives on granules of HIP virtual memory (csrc/tvdn_devmem.hip), which RCCL is handed only after the self-check has also
    # passed with its states on granules -- else this rank's state is a plain hipMalloc block, loudly as well.
    overlap_ok, granules_ok = True, None
    if world > 1 and backend_factory is None and dist.get_backend(group) == "nccl":
        import warnings
        ok = exchange_verified(group, device)
        if not ok["blocking"]:
            raise RuntimeError(f"halo exchange over {ok['transport']} does not reproduce the single-GPU result: {ok}")
        if not ok["overlap"]:
            warnings.warn(f"overlapped halo exchange failed its self-check on this process group ({ok}); "
                          "using the blocking exchange", RuntimeWarning)
            overlap_ok = False
        g = ok.get("granules")
        if g is not None:
            granules_ok = bool(g["blocking"])
            if not granules_ok:
                warnings.warn(f"halo exchange straight out of / into device memory on granules failed its self-check ({g}); "
                              "this run keeps its state on plain hipMalloc blocks (as TVDN_VMM=0 would)", RuntimeWarning)
            elif not g["overlap"]:
                overlap_ok = False
    be = (backend_factory or (lambda l: HipBackend(l, dtype, FISTA, device=device, max_iters=n,
                                                   granules=None if granules_ok is not False else False)))(lay)
    be.set_params(1.0 / lam, (lam / mu).astype(dtype))
    # own rows in, halo rows from the neighbours
    block = torch.zeros(lay.local_shape, dtype=torch.float32 if dtype == np.float32 else torch.float64)
    block[lay.row_lo:lay.row_hi] = my_rows.cpu() if is_t else torch.from_numpy(np.ascontiguousarray(my_rows))
    be.set_input(block)
    run = SlabRunner(be, group)
    run.overlap = overlap_ok
    run.exchange_halos()
    dt = dtype.type

    def on_iter(slot):
        if stopping_relative_change is None:
            return False
        run.finish()
        s = be.sums_tensor()[slot].clone()
        dist.all_reduce(s, group=group)          # global criterion: every rank takes the same decision
        s = s.cpu().numpy()
        with np.errstate(divide="ignore", invalid="ignore"):
            return bool(dt(dt(s[1]) / dt(s[2])) < stopping_relative_change)

    if FISTA:
        run.run(n_f, 0, on_iter)
        run.iter = n_f
    if unacc:
        run.run(0, n_p, on_iter)
    gs = run.global_sums()
    # (device memory comes home through the library's pinned lanes, never through the runtime's path for pageable memory, which pins
    # the destination in place and caches the pin beyond the array's life: profiles/r06_abort_found.txt; test backends are host-side)
    sums = (_to_host(gs, np.float64, device) if gs.is_cuda else gs.numpy())[:n]
    ran = np.zeros(n, dtype=bool)
    ran[run.ran] = True
    b_norm = np.where(ran, sums[:, 0], 0.0).astype(dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        delta = np.where(ran, sums[:, 1].astype(dtype) / sums[:, 2].astype(dtype), dt(0)).astype(dtype)
    own = be.recon_tensor()[lay.row_lo:lay.row_hi]
    if is_t and my_rows.is_cuda:
        return own, b_norm, delta
    return (_to_host(own, dtype, device) if own.is_cuda else own.numpy()), b_norm, delta


def _to_host(t, dtype, device):
    """A device tensor as a fresh NumPy array, through the library's pinned lanes (csrc/tvdn_hostio.hip)."""
    from . import _lib
    torch.cuda.current_stream(int(device) if not isinstance(device, torch.device) else device).synchronize()
    return _lib.copy_to_host(t.contiguous(), dtype)


class _RankHooks:
    """What crosses process boundaries in a multi-process streamed run (tvdn_slab_io): the library calls these on the calling
    thread, at the same points of its schedule on every rank.  gloo moves host memory directly; RCCL stages rows through HBM."""

    def __init__(self, dist, group, rank, world, device, periodic):
        self.dist, self.group, self.rank, self.world = dist, group, rank, world
        self.via_dev = dist.get_backend(group) != "gloo"
        self.cuda = torch.device("cuda", device)
        self.peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
        ring = periodic and world > 1
        self.left = (rank - 1) % world if (rank > 0 or ring) else None
        self.right = (rank + 1) % world if (rank < world - 1 or ring) else None
        self._relay_work = self._relay_buf = None
        self.error = None

    @staticmethod
    def _view(ptr, rows, row_bytes):
        import ctypes as C
        raw = np.ctypeslib.as_array(C.cast(C.c_void_p(ptr), C.POINTER(C.c_uint8)), shape=(int(rows) * int(row_bytes),))
        return torch.from_numpy(raw).view(int(rows), int(row_bytes))

    def _snd(self, t):
        return t.to(self.cuda) if self.via_dev else t.contiguous()

    STAGE_BYTES = 512 << 20      # RCCL: device memory per direction that a shift may stage rows through

    def _shift(self, views, take, put, to, frm, tag0):
        """Everybody sends rows `take` of every array to `to` and receives rows `put` from `frm` (either may be None).
        Over gloo the page-locked host rows travel as they are, all arrays in one batch.  Over RCCL they are staged through HBM,
        and at the moment of a swap the library holds its rings and as many resident rows as fit 85 % of the free HBM: the rows
        go array by array in chunks of at most STAGE_BYTES through TWO reusable device buffers (BASELINE configs[4]: k = 16
        rows of 256 MiB x 9 arrays would otherwise ask torch for 72 GiB that are no longer there).  Both sides cut
        the same chunks -- the counts follow from (rows, row bytes) alone -- so the messages pair up."""
        dist = self.dist
        if not self.via_dev:
            ops = []
            for i, t in enumerate(views):
                if to is not None:
                    ops.append(dist.P2POp(dist.isend, t[take].contiguous(), self.peer(to), self.group, tag=tag0 + 4 * i))
                if frm is not None:
                    ops.append(dist.P2POp(dist.irecv, t[put], self.peer(frm), self.group, tag=tag0 + 4 * i))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            return
        n_rows = (take.stop - take.start)
        row_bytes = int(views[0].shape[1]) if views else 0
        if not views or n_rows <= 0 or (to is None and frm is None):
            return
        chunk = max(1, min(n_rows, self.STAGE_BYTES // max(1, row_bytes)))
        if getattr(self, "_stage", None) is None or self._stage[0].shape != (chunk, row_bytes):
            self._stage = [torch.empty((chunk, row_bytes), dtype=torch.uint8, device=self.cuda) for _ in range(2)]
        snd, rcv = self._stage
        for i, t in enumerate(views):
            for c0 in range(0, n_rows, chunk):
                c = min(chunk, n_rows - c0)
                ops = []
                if to is not None:
                    snd[:c].copy_(t[take.start + c0:take.start + c0 + c])
                    ops.append(dist.P2POp(dist.isend, snd[:c], self.peer(to), self.group, tag=tag0 + 4 * i))
                if frm is not None:
                    ops.append(dist.P2POp(dist.irecv, rcv[:c], self.peer(frm), self.group, tag=tag0 + 4 * i))
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                if frm is not None:
                    t[put.start + c0:put.start + c0 + c].copy_(rcv[:c])

    def exchange(self, _user, n, arrays, rows_per, lo, hi, depth, row_bytes):
        try:
            views = [self._view(arrays[i], rows_per, row_bytes) for i in range(n)]
            d = int(depth)
            # two uniform shifts (no chain of dependent messages): up -- my highest own rows become the right neighbour's low
            # halo rows --, then down
            self._shift(views, slice(hi - d, hi), slice(lo - d, lo), self.right, self.left, 2)
            self._shift(views, slice(lo, lo + d), slice(hi, hi + d), self.left, self.right, 1)
            return 0
        except Exception as e:          # never through the C frames: the library turns the status into its own error
            self.error = e
            return 1

    def allreduce(self, _user, sums3):
        try:
            t = torch.tensor([sums3[0], sums3[1], sums3[2]], dtype=torch.float64)
            if self.via_dev:
                t = t.to(self.cuda)
            self.dist.all_reduce(t, group=self.group)
            t = t.cpu()
            for j in range(3):
                sums3[j] = float(t[j])
            return 0
        except Exception as e:
            self.error = e
            return 1

    def relay_row0(self, _user, send, planes, n, row_bytes):
        """Row 0 of every level of a pass, from the rank that owns it to every other rank (a broadcast every rank joins once
        per pass; the owner does not wait for the others, which arrive late in their passes)."""
        try:
            v = self._view(planes, n, row_bytes)
            if send:
                if self._relay_work is not None:
                    self._relay_work.wait()
                self._relay_buf = self._snd(v).clone()
                self._relay_work = self.dist.broadcast(self._relay_buf, self.peer(0), group=self.group, async_op=True)
            else:
                buf = torch.empty(v.shape, dtype=v.dtype, device=self.cuda) if self.via_dev else v
                self.dist.broadcast(buf, self.peer(0), group=self.group)
                if self.via_dev:
                    v.copy_(buf)
            return 0
        except Exception as e:
            self.error = e
            return 1

    def finish(self):
        if self._relay_work is not None:
            self._relay_work.wait()
            self._relay_work = self._relay_buf = None


def _denoise_slabs_staged(my_rows, lay, dtype, mu, lam, FISTA, unacc, n_f, n_p, stop, group, device, staged, rank, world,
                          exact_wrap=False):
    """Every rank streams ITS slab through its GPU from its own page-locked host memory with the library's streamed loop
    (tvdn_run with tvdn_run_args.slab, csrc/tvdn_stream.hip run_streamed_rank); between the passes the hooks above refresh
    the k halo rows of recon and of the accumulator state from the neighbours."""
    import ctypes as C
    import torch.distributed as dist
    from . import _lib
    rows, k = max(1, int(staged[0])), max(1, int(staged[1]))
    # rows of the slab that keep their state in HBM between the passes: staged = (rows, k, n) caps them, (rows, k) = as many
    # interior rows as fit beside the rings (none of the k a neighbour reads at a shared face; planner.plan_run says how many)
    resident = int(staged[2]) if len(staged) > 2 and not isinstance(staged[2], str) else -1
    n = n_f + n_p
    own = my_rows.cpu().numpy() if isinstance(my_rows, torch.Tensor) else np.ascontiguousarray(my_rows)
    nd = own.ndim
    if world == 1:          # nothing crosses a process boundary: the one-device streamed run
        from .driver import _run_device_list
        return _run_device_list([int(device)], own, 1.0 / lam, (lam / mu).astype(dtype), n_f if FISTA else 0,
                                n_p if unacc else 0, stop, None, lay.bc_mode, True, stream=(rows, k))[:3]
    k = min(k, lay.shape[0] // world)       # a pass reads k rows of the neighbour's state: at most the smallest slab's rows
    hooks = _RankHooks(dist, group, rank, world, device, lay.bc_mode == 0)
    io = _lib.SlabIO(global_rows=int(lay.shape[0]), row0=int(lay.g0), rank=int(rank), world=int(world),
                     first_row_nonfinite=int(bool(exact_wrap)))
    cb = (_lib.SLAB_EXCHANGE(hooks.exchange), _lib.SLAB_ALLREDUCE(hooks.allreduce), _lib.SLAB_RELAY(hooks.relay_row0))
    io.exchange, io.allreduce, io.relay_row0 = cb
    a = _lib.RunArgs(dtype=_lib.dtype_code(dtype), ndim=nd, bc_mode=int(lay.bc_mode), device=int(device),
                     n_fista=n_f if FISTA else 0, n_plain=n_p if unacc else 0, use_stop=int(stop is not None),
                     stop=float(stop or 0.0), stream_rows=rows, stream_k=k, stream_resident=resident)
    for i, v in enumerate(own.shape):
        a.shape[i] = int(v)
    lam_inv, lam_mu = 1.0 / lam, (lam / mu).astype(dtype)
    for q in range(nd):
        a.clip[q], a.lambda_mu[q] = float(lam_inv[q]), float(lam_mu[q])
    recon = np.empty_like(own)
    sums = np.zeros((max(n, 1), 3))
    phases = (C.c_int32 * 2)(0, 0)
    a.data, a.recon_out, a.sums_out = own.ctypes.data, recon.ctypes.data, sums.ctypes.data
    a.phase_iters = C.addressof(phases)
    a.slab = C.pointer(io)
    # Before any rank page-locks anything: what will the ranks of each HOST page-lock together?  (The guard inside the library
    # sees one rank; two ranks that each pass it can exhaust the memory their host -- or their control group -- allows.)
    need, kept = C.c_int64(0), C.c_int64(0)
    rc_need = _lib.lib().tvdn_slab_host_need(C.byref(a), C.byref(need), C.byref(kept))
    # ... plus what is not page-locked but lives in the same memory: the slab and its result as this rank holds them, and the rows
    # of one swap in flight (gloo moves them through host memory: k rows of every swapped array, sent and received)
    row_bytes = int(np.prod(own.shape[1:])) * own.dtype.itemsize
    beside = own.nbytes + recon.nbytes + (0 if hooks.via_dev else 2 * k * row_bytes * (1 + nd * (2 if n_f and FISTA else 1)))
    _check_hosts_hold_the_slabs(dist, group, world, int(need.value) + beside if rc_need == 0 else -1,
                                None if rc_need == 0 else _lib.lib().tvdn_last_error().decode())
    rc = _lib.lib().tvdn_run(C.byref(a))
    hooks.finish()
    if hooks.error is not None:
        raise hooks.error
    _lib.check(rc)
    t = torch.from_numpy(sums[:n].copy())
    if n:
        if hooks.via_dev:
            t = t.to(hooks.cuda)
        dist.all_reduce(t, group=group)          # the traces are global and identical on every rank
        t = t.cpu()
    sums = t.numpy()
    done = np.zeros(n, dtype=bool)
    done[:phases[0]] = True
    done[(n_f if FISTA else 0):(n_f if FISTA else 0) + phases[1]] = True
    dt = dtype.type
    b_norm = np.where(done, sums[:, 0], 0.0).astype(dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        delta = np.where(done, sums[:, 1].astype(dtype) / sums[:, 2].astype(dtype), dt(0)).astype(dtype)
    return recon, b_norm, delta


def _check_hosts_hold_the_slabs(dist, group, world, need_bytes, error=None, host=None, available=None):
    """Collective: every rank contributes what it will page-lock (or its error); the sums per host are held against 80 % of
    what that host has available, and every rank raises the same MemoryError (RuntimeError for a rank's own error) if any
    host would be overdrawn -- nobody is left waiting in a collective for a rank that has bailed out."""
    import socket
    from .planner import HOST_FRACTION, host_available
    mine = {"host": host or socket.gethostname(), "need": int(need_bytes), "error": error,
            "available": available if available is not None else host_available()}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine, group=group)
    errors = [f"rank {r}: {e['error']}" for r, e in enumerate(everyone) if e["error"]]
    if errors:
        raise RuntimeError("a rank cannot size its slab: " + "; ".join(errors))
    per_host = {}
    for e in everyone:
        h = per_host.setdefault(e["host"], {"need": 0, "ranks": 0, "available": e["available"]})
        h["need"] += e["need"]
        h["ranks"] += 1
        if e["available"] is not None:
            h["available"] = e["available"] if h["available"] is None else min(h["available"], e["available"])
    over = [f"{name}: {h['ranks']} ranks x their slabs = {h['need'] / 2 ** 30:.1f} GiB of page-locked host memory, "
            f"{HOST_FRACTION:.0%} of the {h['available'] / 2 ** 30:.1f} GiB available is {HOST_FRACTION * h['available'] / 2 ** 30:.1f} GiB"
            for name, h in sorted(per_host.items()) if h["available"] is not None and h["need"] > HOST_FRACTION * h["available"]]
    if over:
        raise MemoryError("the slabs do not fit the host memory of " + "; ".join(over) + ": fewer ranks per host, a shallower k "
                          "(2 k halo rows per array and rank), or more rows resident in HBM (planner.plan_run(..., host_bytes=...))")


_VERIFIED = {}


def exchange_verified(group=None, device=None) -> dict:
    """selfcheck_exchange, run once per process group and remembered: the check on plain device memory, and under
    "granules" the same check with its states on granules of HIP virtual memory (what a slab of 2 GiB or more lives on;
    None when granules are off anyway: TVDN_VMM=0, a runtime without virtual-memory management, a tripped remap canary)."""
    key = id(group) if group is not None else 0
    if key not in _VERIFIED:
        res = selfcheck_exchange(group=group, device=device)
        res["granules"] = None
        if res["blocking"] and os.environ.get("TVDN_VMM", "1") != "0":
            g = selfcheck_exchange(group=group, device=device, on_granules=True)
            res["granules"] = {k: g[k] for k in ("overlap", "blocking", "error", "state_mem")}
        _VERIFIED[key] = res
    return _VERIFIED[key]


def selfcheck_exchange(group=None, device=None, iterations: int = 6, dtype=np.float32, plane=(6, 16, 32), on_granules: bool = False):
    """Pre-flight check of the multi-GPU exchange on THIS process group: a small cube (4 rows per rank) is
    denoised three ways -- one slab on this rank's own GPU (no communication), the slab runner with the halo
    exchange overlapped under the interior sweep (`step_overlapped`), and the slab runner with a blocking
    exchange after every sweep -- and this rank's rows must come out bit-identical in all three.  The verdicts are
    combined over all ranks (minimum), so every rank returns the same dict:
        {"overlap": bool, "blocking": bool, "transport": "rccl" | "gloo", "error": str | None}
    A failing transport shows up as False (an exception is caught and reported), never as a silent fallback.

    `on_granules=True` puts the slab runs' states on granules of HIP virtual memory (csrc/tvdn_devmem.hip; the threshold is
    lowered to 0 for the duration), the memory a slab of 2 GiB or more lives on and that a transport must prove it can send out
    of and receive into.  "state_mem" says what the states really were ("granules" / "plain", the worst over all ranks); a
    check that was asked for granules and ran on anything else counts as FAILED, not as green."""
    import torch.distributed as dist
    from . import synth
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if device is None:
        device = torch.cuda.current_device()
    dt = np.dtype(dtype)
    shape = (4 * world,) + tuple(plane)
    nd = len(shape)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    x = synth.cube(shape, seed=4242, dtype=dt) + dt.type(0.25)

    mems = []

    def run(lay, overlap, grp, slab=True):
        be = HipBackend(lay, dt, True, device=device, max_iters=iterations, granules=True if (on_granules and slab) else None)
        if slab:
            mems.append(be.state_mem)
        be.set_params(1.0 / lam, (lam / mu).astype(dt))
        be.set_input(x[lay.local_rows_global()])
        r = SlabRunner(be, grp)
        r.overlap = overlap
        r.run(iterations, 0)
        torch.cuda.synchronize(device)
        return be.recon_tensor()[lay.row_lo:lay.row_hi].clone(), r.transport

    res = {"overlap": False, "blocking": False, "transport": None, "error": None, "world": world, "ranks_seen": None,
           "distinct_gpus": None}
    # who is really there: every rank adds 1 through the data-plane group itself, and names its GPU (PCI bus id) through
    # whatever group exists -- eight ranks that all landed on one GPU would pass every bit test and measure nothing
    try:
        one = torch.ones(1, dtype=torch.int32)
        if dist.get_backend(group) == "nccl":
            one = one.to(torch.device("cuda", device))
        dist.all_reduce(one, group=group)
        res["ranks_seen"] = int(one.item())
        try:
            bus = str(torch.cuda.get_device_properties(device).pci_bus_id) + ":" + str(torch.cuda.get_device_properties(device).pci_device_id)
        except Exception:
            bus = f"index{device}"
        names = [None] * world
        dist.all_gather_object(names, f"{os.uname().nodename}/{bus}/{device}", group=group if dist.get_backend(group) != "nccl" else None)
        res["distinct_gpus"] = len(set(n for n in names if n))
    except Exception as e:
        res["error"] = f"census: {e!r}"
    lay = SlabLayout(shape, rank, world, 2)
    want, _ = run(SlabLayout(shape, 0, 1, 2), False, None, slab=False)
    want = want[lay.g0:lay.g1]
    for key, overlap in (("blocking", False), ("overlap", True)):
        try:
            got, res["transport"] = run(lay, overlap, group)
            res[key] = bool(torch.equal(got.view(torch.int32 if dt == np.float32 else torch.int64),
                                        want.view(torch.int32 if dt == np.float32 else torch.int64)))
        except Exception as e:  # report, do not hide
            res["error"] = f"{key}: {e!r}"
    on_gr = int(bool(mems) and all(m == "granules" for m in mems))
    if on_granules and not on_gr:
        res["overlap"] = res["blocking"] = False
        res["error"] = res["error"] or f"the self-check's states were asked for on granules and came as {sorted(set(mems)) or 'nothing'}"
    flags = torch.tensor([int(res["overlap"]), int(res["blocking"]), on_gr], dtype=torch.int32)
    if dist.get_backend(group) == "nccl":
        flags = flags.to(torch.device("cuda", device))
    dist.all_reduce(flags, op=dist.ReduceOp.MIN, group=group)
    res["overlap"], res["blocking"] = bool(flags[0].item()), bool(flags[1].item())
    res["state_mem"] = "granules" if int(flags[2].item()) else "plain"
    return res


__all__ = ["denoise_slabs", "slab_rows", "selfcheck_exchange", "exchange_verified"]
