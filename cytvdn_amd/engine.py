"""Device-resident iteration engine: one slab of the datacube per GPU.

The reference's only multi-process path tiles real space in 2-D over MPI ranks and patches 1-row
halos with mpi4py messages (reference cyTVDN/mpi.py:131-210, :314-434).  Here the same idea is
expressed MI355X-first: axis 0 is cut into contiguous slabs, one per GPU, so that a halo row is a
single contiguous block that RCCL sends straight out of / into the recon buffer (no packing), and
only recon is ever exchanged (accumulators of the halo row are recomputed locally).

Non-finite data.  Under the Jia-Zhao boundary condition the axis-0 accumulator of global row 0 is identically zero
for finite data (clip((r - r) + 0)), which is what lets the last slab close the periodic wrap of the reconstruction
update with a constant (TVDN_EDGE_ZERO) instead of a message from rank 0.  If row 0 holds an Inf or a NaN, upstream
(and the single-slab run here, which wraps for real) gets NaN from Inf - Inf in that accumulator and propagates it
into the LAST row.  `SlabLayout(..., wrap_row=True)` reproduces that: rank 0 then also sends its first row to the last
rank every iteration, which forms the wrapped accumulator from it as upstream does (TVDN_EDGE_WRAP).  `denoise_slabs`,
`denoise3D/4D` and `tvdn_run` (resident and streamed) switch it on by themselves when the first row of the input is
not finite; `bench.py` and finite data never pay for the extra message.  Staged slabs across ranks
(`denoise_slabs(staged=...)`) relay row 0 of every level of a pass from rank 0 to the last rank (tvdn.h, tvdn_slab_io
relay_row0; csrc/tvdn_stream.hip), with or without a stopping rule.

Pieces
------
SlabLayout   pure bookkeeping: which global rows a rank owns, which halo rows it keeps, how its two
             edges are treated by the fused sweep (tvdn.h TVDN_EDGE_*), who its neighbours are.
HipBackend   the product backend: state in HBM (torch tensors are only the allocator), one
             `tvdn_iterate_fused` launch per iteration (double-buffered state, see csrc/tvdn_fused.hip).
SlabRunner   drives a backend: FISTA schedule (float64 on the host as upstream, cyTVDN.py:153-156),
             per-iteration halo exchange over torch.distributed (RCCL on GPUs; any backend object with
             the same five methods can be driven, which is how tests rehearse the protocol with gloo).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib


# accumulator state representation used by HipBackend unless told otherwise (see its docstring)
DEFAULT_STATE = os.environ.get("TVDN_STATE", "compact")
ARRAY_SKEW = 4096   # bytes by which consecutive state arrays are staggered inside the state allocation


def vmm_min_bytes() -> int:
    """States from this size on are composed from physical granules: csrc/tvdn_devmem.hip's own threshold (2 GiB;
    TVDN_VMM_MIN_MIB overrides both, tests use it to put small cubes on granules)."""
    return int(os.environ.get("TVDN_VMM_MIN_MIB", "2048")) << 20


class _vmm_threshold:
    """For the duration of one allocation: TVDN_VMM_MIN_MIB = `mib` (None: leave it alone).  The library reads the variable
    when it allocates; 0 puts every block on granules (csrc/tvdn_devmem.hip env_mib)."""

    def __init__(self, mib):
        self.mib = mib

    def __enter__(self):
        if self.mib is not None:
            self.prev = os.environ.get("TVDN_VMM_MIN_MIB")
            os.environ["TVDN_VMM_MIN_MIB"] = str(int(self.mib))

    def __exit__(self, *exc):
        if self.mib is not None:
            if self.prev is None:
                os.environ.pop("TVDN_VMM_MIN_MIB", None)
            else:
                os.environ["TVDN_VMM_MIN_MIB"] = self.prev
        return False


def fista_ratios(n: int) -> np.ndarray:
    """(tk-1)/tk_new for iterations 0..n-1, float64 on the host (reference cyTVDN.py:153-156): the library's own
    recurrence (_lib.fista_ratios), so that this schedule is defined once for every engine."""
    return _lib.fista_ratios(n)


@dataclass(frozen=True)
class SlabLayout:
    """Rows of axis 0 owned by `rank` out of `world`, plus halo bookkeeping."""
    shape: tuple          # GLOBAL shape
    rank: int
    world: int
    bc_mode: int
    bounds: tuple = None  # optional explicit partition: world+1 increasing row indices, 0 .. shape[0]
    wrap_row: bool = False  # Jia-Zhao, several slabs: the last slab also keeps the CURRENT recon of global row 0 (sent by
    #                         rank 0 every iteration) and forms the wrapped axis-0 accumulator from it as upstream does
    #                         (TVDN_EDGE_WRAP) instead of taking it as zero: exact when row 0 holds Inf/NaN as well

    def __post_init__(self):
        if self.world < 1 or not (0 <= self.rank < self.world):
            raise ValueError("bad rank/world")
        if self.shape[0] < self.world:
            raise ValueError(f"axis 0 ({self.shape[0]} rows) cannot be cut into {self.world} slabs")
        if self.bc_mode not in (0, 2):
            raise NotImplementedError("slab engine supports BC_mode 0 (periodic) and 2 (Jia-Zhao)")
        if self.bounds is not None:
            b = tuple(int(v) for v in self.bounds)
            if len(b) != self.world + 1 or b[0] != 0 or b[-1] != self.shape[0] or any(y <= x for x, y in zip(b, b[1:])):
                raise ValueError("bounds must be world+1 strictly increasing row indices from 0 to shape[0]")
            object.__setattr__(self, "bounds", b)

    # global rows [g0, g1) owned by this rank (balanced split unless `bounds` says otherwise)
    @property
    def g0(self) -> int:
        if self.bounds is not None:
            return self.bounds[self.rank]
        return (self.rank * self.shape[0]) // self.world

    @property
    def g1(self) -> int:
        if self.bounds is not None:
            return self.bounds[self.rank + 1]
        return ((self.rank + 1) * self.shape[0]) // self.world

    @property
    def ring(self) -> bool:
        return self.world > 1 and self.bc_mode == 0

    @property
    def halo_lo(self) -> int:
        return 1 if self.world > 1 and (self.rank > 0 or self.ring) else 0

    @property
    def wraps(self) -> bool:
        return bool(self.wrap_row) and self.world > 1 and self.bc_mode == 2

    @property
    def halo_hi(self) -> int:
        return 1 if self.world > 1 and (self.rank < self.world - 1 or self.ring or self.wraps) else 0

    @property
    def own_rows(self) -> int:
        return self.g1 - self.g0

    @property
    def local_shape(self) -> tuple:
        return (self.halo_lo + self.own_rows + self.halo_hi,) + tuple(self.shape[1:])

    @property
    def row_lo(self) -> int:
        return self.halo_lo

    @property
    def row_hi(self) -> int:
        return self.halo_lo + self.own_rows

    @property
    def lo_mode(self) -> int:
        return _lib.EDGE_HALO if self.halo_lo else _lib.EDGE_BC

    @property
    def hi_mode(self) -> int:
        if self.wraps and self.rank == self.world - 1:
            return _lib.EDGE_WRAP
        if self.halo_hi:
            return _lib.EDGE_HALO
        return _lib.EDGE_BC if self.world == 1 else _lib.EDGE_ZERO

    @property
    def left(self):
        """Rank that owns the row below g0 (None at a non-periodic global edge)."""
        if not self.halo_lo:
            return None
        return (self.rank - 1) % self.world

    @property
    def right(self):
        if not self.halo_hi or (self.rank == self.world - 1 and not self.ring):
            return None
        return (self.rank + 1) % self.world

    @property
    def wrap_to(self):
        """Rank that keeps a copy of my first row (rank 0 of a `wrap_row` layout -> the last rank)."""
        return self.world - 1 if self.wraps and self.rank == 0 else None

    @property
    def wrap_from(self):
        return 0 if self.wraps and self.rank == self.world - 1 else None

    def local_rows_global(self) -> np.ndarray:
        """Global row index held by each local row (halo rows included, periodic wrap applied)."""
        rows = np.arange(self.g0 - self.halo_lo, self.g1 + self.halo_hi)
        return rows % self.shape[0]


class HipBackend:
    """State of one slab in HBM + the fused sweep.  torch is the allocator, nothing more.

    state="compact" (default): FISTA keeps, per axis, three rotating arrays d_k+1 / d_k / d_k-1 and
    rebuilds b_k = d_k + tk_prev*(d_k - d_k-1) on the fly (tvdn.h TVDN_ITER_FISTA_D): 15 arrays and
    15 array passes per 4-D iteration.  Unaccelerated iterations ping-pong b in two of those arrays.
    state="reference": the reference's own (b, d) pair per axis, double-buffered (TVDN_ITER_FISTA):
    19 arrays, 19 passes; kept for cross-checks."""

    supports_partial_sweeps = True

    def __init__(self, layout: SlabLayout, dtype, fista: bool, device: int = 0, max_iters: int = 1,
                 state: str = None, private_ctx: bool = False, slab=None, granules=None):
        """`slab` (measurement, tools/layout_probe.py): a 1-D device tensor of the data dtype to carve the arrays from
        instead of a fresh allocation -- several layouts of the state timed on the very same pages.
        `granules`: None = the library's rule (states of 2 GiB and more on granules of HIP virtual memory, csrc/tvdn_devmem.hip);
        False = a plain block whatever the size (a transport that failed its self-check on granules: distributed.denoise_slabs);
        True = granules whatever the size (the self-check itself) -- `state_mem` says what it became."""
        state = DEFAULT_STATE if state is None else state
        if state not in ("compact", "reference"):
            raise ValueError("state must be 'compact' or 'reference'")
        self.layout = layout
        self.dtype = np.dtype(dtype)
        self.code = _lib.dtype_code(self.dtype)
        self.fista = bool(fista)
        self.state = state
        self.device = int(device)
        self.nd = len(layout.shape)
        if self.nd not in (3, 4):
            raise TypeError("No matching signature found")
        # raises without a GPU: no CPU fallback.  A private context = own reduction scratch, needed when two
        # backends are driven from two streams or threads at once (the audition, driver.py)
        self._private_ctx = bool(private_ctx)
        self.ctx = _lib.new_ctx(self.device) if private_ctx else _lib.ctx(self.device)
        tdt = torch.float32 if self.dtype == np.float32 else torch.float64
        dev = torch.device("cuda", self.device)
        ls = layout.local_shape
        # ONE allocation for the whole state, carved into arrays: one hipMalloc instead of 11-19, and one fill for
        # every array except the two that `set_input` overwrites anyway (orig and the current recon).  Arrays start
        # 256-byte aligned and staggered by ARRAY_SKEW bytes each: with every array at the same offset modulo a large
        # power of two the 15 streams of a sweep compete for the same cache sets / DRAM banks at the same moment
        # (interleaved A/B on config 2, four rounds: 11.42/12.25/12.80/12.79 ms unstaggered against
        # 11.26/11.30/12.08/11.53 ms with 4 KiB; 2 KiB in between).  TVDN_ARRAY_SKEW overrides (multiple of 256).
        n_el = int(np.prod(ls))
        item = self.dtype.itemsize
        skew = int(os.environ.get("TVDN_ARRAY_SKEW", str(ARRAY_SKEW)))
        stride_el = (-(-(n_el * item) // 256) * 256 + skew) // item
        per_axis = (3 if fista else 2) if state == "compact" else (4 if fista else 2)
        n_arr = 3 + self.nd * per_axis
        it = iter(range(n_arr))
        if os.environ.get("TVDN_ALLOC", "one") == "separate":
            # measurement knob: one allocation PER ARRAY (each then gets its own run of physical pages, power-of-two sized
            # arrays their own naturally aligned blocks), staggered inside its allocation by the same ARRAY_SKEW
            self._slab = None
            self._parts = []
            pad_el = (n_arr * skew) // item + 64

            def arr():
                i = next(it)
                t = torch.empty(n_el + pad_el, dtype=tdt, device=dev)
                if i < n_arr - 2:
                    t.zero_()
                self._parts.append(t)
                off = (i * skew) // item
                return t[off:off + n_el].view(ls)
        else:
            if slab is not None:
                if slab.dtype != tdt or slab.numel() < n_arr * stride_el:
                    raise ValueError(f"slab must hold {n_arr * stride_el} elements of {tdt}")
                self._slab = slab[:n_arr * stride_el]
            elif granules is not False and os.environ.get("TVDN_VMM", "1") != "0" \
                    and (granules is True or n_arr * stride_el * item >= vmm_min_bytes()):
                # a big state: the library's allocator (granules; how fast the sweep runs on a hipMalloc block of this size
                # is decided by where it landed -- csrc/tvdn_devmem.hip).  The block outlives the tensor views below.
                with _vmm_threshold(0 if granules is True else None):
                    self._block = _lib.DeviceBlock(n_arr * stride_el * item, self.device)
                self._slab = self._block.tensor(tdt)
            else:
                self._slab = torch.empty(n_arr * stride_el, dtype=tdt, device=dev)
            self._slab[:(n_arr - 2) * stride_el].zero_()

            def arr():
                i = next(it)
                return self._slab[i * stride_el:i * stride_el + n_el].view(ls)

        if state == "reference":
            self.b = [[arr(), arr()] for _ in range(self.nd)]
            self.d = [[arr(), arr()] for _ in range(self.nd)] if fista else None
        else:
            # S[q][k]: k-th rotating array of axis q (3 with FISTA: d_k, d_k-1, next; 2 without: b, next)
            self.S = [[arr() for _ in range(per_axis)] for _ in range(self.nd)]
        r1 = arr()                                           # zeroed: its halo rows are read before any exchange fills them
        self.orig = arr()
        self.recon = [arr(), r1]
        # Which array plays which role (recon[cur]; d_k / d_k-1 / next while the state is in d-form, b / next in b-form)
        # lives in a tvdn_many_args that the library binds and rotates (tvdn_roles_bind / tvdn_roles_advance): the
        # same code tvdn_iterate_many and tvdn_run execute.  All-zero (d_k, d_k-1) == all-zero b.
        self._roles = _lib.ManyArgs()
        m = self._roles
        m.recon[0], m.recon[1] = self.recon[0].data_ptr(), self.recon[1].data_ptr()
        if state == "compact":
            for q in range(self.nd):
                for k, t in enumerate(self.S[q]):
                    m.S[q][k] = t.data_ptr()
        m.cur, m.i_d, m.i_prev, m.i_out, m.i_b, m.i_bout = 0, 0, 1, 2, 0, 1
        m.d_form, m.tk_prev = int(bool(fista)), 0.0
        self.sums = torch.zeros((max(int(max_iters), 1), 3), dtype=torch.float64, device=dev)
        self._mode = None
        self._args = _lib.IterArgs()
        a = self._args
        a.dtype, a.ndim = self.code, self.nd
        for i, s in enumerate(ls):
            a.shape[i] = int(s)
        a.row_lo, a.row_hi = layout.row_lo, layout.row_hi
        a.lo_mode, a.hi_mode, a.bc_mode = layout.lo_mode, layout.hi_mode, layout.bc_mode
        a.orig = self.orig.data_ptr()

    @property
    def state_mem(self) -> str:
        """What the state's block is made of: "granules" (tvdn_mem_alloc composed it), "plain" (one hipMalloc block: torch's
        allocator, or the library's below its threshold / without virtual-memory management) or "caller" (`slab=`)."""
        b = getattr(self, "_block", None)
        if b is not None:
            return "granules" if b.kind == _lib.MEM_GRANULES else "plain"
        return "plain"

    def release(self):
        """Give the state's device memory back now (the views into it must not be used afterwards)."""
        for name in ("S", "b", "d", "recon", "orig", "_slab", "_parts"):
            if hasattr(self, name):
                setattr(self, name, None)
        b = getattr(self, "_block", None)
        if b is not None:
            torch.cuda.current_stream(self.device).synchronize()
            b.free()
            self._block = None

    def __del__(self):
        try:
            if getattr(self, "_private_ctx", False) and self.ctx:
                _lib.lib().tvdn_ctx_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass
        try:
            self.release()
        except Exception:
            pass

    # role bookkeeping: views of the library-side struct
    cur = property(lambda self: int(self._roles.cur), lambda self, v: setattr(self._roles, "cur", int(v)))
    i_d = property(lambda self: int(self._roles.i_d))
    i_prev = property(lambda self: int(self._roles.i_prev))
    i_out = property(lambda self: int(self._roles.i_out))
    i_b = property(lambda self: int(self._roles.i_b))
    i_bout = property(lambda self: int(self._roles.i_bout))
    d_form = property(lambda self: bool(self._roles.d_form))
    tk_prev = property(lambda self: float(self._roles.tk_prev))

    # -- the five methods SlabRunner needs ------------------------------------------------------
    def set_params(self, clip, lam_mu):
        for q in range(self.nd):
            self._args.clip[q] = float(clip[q])
            self._args.lambda_mu[q] = float(lam_mu[q])

    def set_input(self, local_block):
        """local_block: torch tensor or NumPy array of layout.local_shape (halo rows included)."""
        if isinstance(local_block, torch.Tensor) and local_block.is_cuda:
            self.orig.copy_(local_block, non_blocking=False)
        else:
            # Host memory -- NumPy or a CPU tensor -- always through the library's own pinned lanes (tvdn_hostio.hip): PCIe speed
            # from pageable memory, and the runtime never gets to pin the caller's pages in place (its cache of such pins outlives
            # the memory: the GPU fault of profiles/r06_abort_found.txt).
            h = local_block.detach().numpy() if isinstance(local_block, torch.Tensor) else np.asarray(local_block)
            if tuple(h.shape) != tuple(self.orig.shape):
                raise ValueError(f"set_input: block of shape {tuple(h.shape)}, the slab holds {tuple(self.orig.shape)}")
            torch.cuda.current_stream(self.device).synchronize()
            _lib.copy_to_device(np.ascontiguousarray(h, dtype=self.dtype), self.orig)
        self.recon[self.cur].copy_(self.orig)

    def recon_to_host(self):
        """The own rows of the current reconstruction as a fresh NumPy array."""
        lay = self.layout
        torch.cuda.current_stream(self.device).synchronize()
        return _lib.copy_to_host(self.recon_tensor()[lay.row_lo:lay.row_hi], self.dtype)

    def _bind(self, tk_ratio):
        """Point the argument block at the arrays of the iteration about to run."""
        a = self._args
        use_fista = tk_ratio is not None
        if use_fista and not self.fista:
            raise ValueError("backend was allocated without FISTA state")
        if self.state == "compact":
            _lib.check(_lib.lib().tvdn_roles_bind(C.byref(self._roles), int(use_fista), float(tk_ratio or 0.0), C.byref(a)))
        else:
            i, o = self.cur, self.cur ^ 1
            a.tk = float(tk_ratio) if use_fista else 0.0
            a.tk_prev = 0.0
            a.recon_in, a.recon_out = self.recon[i].data_ptr(), self.recon[o].data_ptr()
            for q in range(self.nd):
                a.b_in[q] = a.b_out[q] = a.d_in[q] = a.d_out[q] = a.dprev_in[q] = None
                a.b_in[q], a.b_out[q] = self.b[q][i].data_ptr(), self.b[q][o].data_ptr()
                if use_fista:
                    a.d_in[q], a.d_out[q] = self.d[q][i].data_ptr(), self.d[q][o].data_ptr()
            a.mode = _lib.ITER_FISTA if use_fista else _lib.ITER_PLAIN
        self._mode = (int(a.mode), tk_ratio)

    def step(self, tk_ratio, slot: int, rows=None, accumulate: bool = False):
        """One iteration over the own rows, or over the sub-range `rows` = (lo, hi) of them.  The state
        arrays rotate when the whole range has been advanced: automatically for a full sweep, by
        `flip()` after a set of partial sweeps."""
        self._bind(tk_ratio)
        a = self._args
        if rows is None:
            a.sweep_lo, a.sweep_hi = 0, 0
        else:
            a.sweep_lo, a.sweep_hi = int(rows[0]), int(rows[1])
        a.accumulate = 1 if accumulate else 0
        _lib.check(_lib.lib().tvdn_iterate_fused(self.ctx, C.byref(a), C.c_void_p(self.sums[slot].data_ptr()),
                                                 _lib.current_stream(self.device)))
        if rows is None:
            self.flip()

    def run_many(self, ratios, n_plain: int, slot0: int):
        """len(ratios) FISTA iterations then n_plain unaccelerated ones over the own rows, behind ONE library call
        (tvdn_iterate_many): the per-iteration host work -- role rotation, tk, sums slot -- happens in C++ instead of
        ~25 us of Python, which is what a small cube's 10-20 us sweeps are otherwise waiting for.  Compact state only;
        same launches, same bits as `step`."""
        if self.state != "compact":
            raise ValueError("run_many drives the compact state")
        n_f = len(ratios)
        if n_f and not self.fista:
            raise ValueError("backend was allocated without FISTA state")
        m = self._roles
        C.memmove(C.byref(m.base), C.byref(self._args), C.sizeof(_lib.IterArgs))   # the fixed part of the argument block
        r = (C.c_double * max(n_f, 1))(*[float(v) for v in ratios])
        _lib.check(_lib.lib().tvdn_iterate_many(self.ctx, C.byref(m), n_f, r, int(n_plain),
                                                C.c_void_p(self.sums[slot0].data_ptr()), _lib.current_stream(self.device)))

    def flip(self):
        """Make the freshly written arrays current (after a full sweep or a set of partial sweeps)."""
        mode, tk_ratio = self._mode
        if self.state == "compact":
            _lib.check(_lib.lib().tvdn_roles_advance(C.byref(self._roles), int(tk_ratio is not None), float(tk_ratio or 0.0)))
        else:
            self.cur ^= 1

    # -- placement audition ------------------------------------------------------------------------------------------
    def probe_ms(self, sweeps: int = 6) -> float:
        """MEAN of `sweeps` timed sweeps on THIS allocation with an all-zero state (the sweep is branch-free: its time
        does not depend on the values -- profiles/r06_data_dependence.jsonl), after one untimed sweep.  Six by default: the
        arrays rotate through their roles with period 6 (three d arrays per axis x two recon buffers), and which arrays are
        written decides the time of a sweep by +- 1.5 % (11.14 ... 11.47 ms within one cycle of config 2, same file) --
        the minimum of two sweeps, which this returned until round 6, is the best arrangement's time, not the block's:
        the four plain blocks of a bench run probed at 10.93 ... 11.36 ms and the best then RAN at 11.30.
        Leaves the state all-zero with its roles reset, i.e. as freshly constructed; call it before set_input."""
        for q in range(self.nd):
            self._args.clip[q], self._args.lambda_mu[q] = 1.0, 1.0 / 32.0
        self.orig.zero_()
        self.recon[0].zero_()
        L = _lib.lib()
        n = 1 + int(sweeps)
        _lib.check(L.tvdn_ctx_timing_enable(self.ctx, 1))
        try:
            for i in range(n):
                self.step(0.5 if self.fista else None, 0)
            each = (C.c_double * (n + 4))()
            nl = C.c_int64()
            _lib.check(L.tvdn_ctx_timing_read_each(self.ctx, each, n + 4, C.byref(nl)))
        finally:
            _lib.check(L.tvdn_ctx_timing_enable(self.ctx, 0))
        self.sums.zero_()
        if self.state == "compact":
            self.set_form(self.fista, 0.0)
        else:
            self.cur = 0
        return float(np.mean(each[1:nl.value])) if nl.value > 1 else float(each[0])

    @classmethod
    def best_of(cls, candidates: int, layout, dtype, fista, device: int = 0, hbm_fraction: float = 0.8,
                release_losers: bool = True, **kw):
        """(Round 5: states of 2 GiB and more are composed from physical granules, csrc/tvdn_devmem.hip, whose sweep time does not
        depend on the draw -- for them this returns the first candidate, unless TVDN_AUDITION insists.  What follows is the
        story of plain hipMalloc blocks, which remains true of them.)
        The sweep's speed depends on WHERE in HBM its state landed: with identical clocks, the same 60 GiB state of
        BASELINE config 2 sweeps in 11.2, 12.1 or 12.6 ms depending on the physical pages one hipMalloc happened to get
        (three states held at once in one process, timed in turn, each reproducible: profiles/r03_placement_audition_*.jsonl;
        plain per-array streaming is equally fast on all of them, so it is the relation BETWEEN the 15 streams -- DRAM
        bank/row conflicts -- not the regions themselves).  The virtual address says nothing (the same address is fast
        in one allocation and slow in the next) and physical addresses are not visible to a user process, so the engine
        auditions: up to `candidates` states are allocated side by side (as many as fit in `hbm_fraction` of the free
        HBM), each is timed for two sweeps, the fastest is kept and the others are freed.  Costs ~3 sweeps per candidate
        before the run; `denoise3D/4D` do it when the run is long enough to pay for it (driver._audition_candidates).
        The result carries `.audition` = the candidates' probe times in ms (kept one first)."""
        item = np.dtype(dtype).itemsize
        per = int(np.prod(layout.local_shape)) * item
        nd = len(layout.shape)
        n_arr = 3 + nd * ((3 if fista else 2) if kw.get("state", DEFAULT_STATE) == "compact" else (4 if fista else 2))
        need = n_arr * (per + ARRAY_SKEW + 256)
        held, times = [], []
        for _ in range(max(1, int(candidates))):
            if held:
                free, _total = torch.cuda.mem_get_info(device)
                if need > hbm_fraction * free:
                    break
            be = cls(layout, dtype, fista, device=device, **kw)
            held.append(be)
            if be.state_mem == "granules" and os.environ.get("TVDN_AUDITION") is None:
                break     # a state on granules sweeps at the same speed whichever granules it got: nothing to audition
            if candidates > 1:
                times.append(be.probe_ms())
        if len(held) == 1:
            held[0].audition = [round(t, 4) for t in times]
            return held[0]
        torch.cuda.current_stream(device).synchronize()
        best = min(range(len(held)), key=lambda i: times[i])
        keep = held[best]
        keep.audition = [round(times[best], 4)] + [round(t, 4) for i, t in enumerate(times) if i != best]
        del held, be
        if release_losers:
            torch.cuda.empty_cache()   # hand the losers' HBM back to the driver (torch caches freed blocks otherwise)
        # else the losers stay in torch's cache: the next audition of this process finds them there instead of asking the
        # driver again -- a hipMalloc of tens of GiB right after a free of that size stalls for seconds some times
        # (profiles/r03_e2e_pipelined.txt), which a process that denoises cube after cube would pay on every call
        return keep

    def set_form(self, d_form: bool, tk_prev: float):
        """Declare what the state arrays hold after an upload: (d_k, d_k-1) pairs or b."""
        if self.state != "compact":
            raise ValueError("only the compact state has forms")
        m = self._roles
        m.cur, m.i_d, m.i_prev, m.i_out, m.i_b, m.i_bout = 0, 0, 1, 2, 0, 1
        m.d_form, m.tk_prev = int(bool(d_form)), float(tk_prev)

    def recon_next(self) -> torch.Tensor:
        """The buffer the sweeps of the current iteration write into."""
        return self.recon[self.cur ^ 1]

    def recon_tensor(self) -> torch.Tensor:
        return self.recon[self.cur]

    def sums_tensor(self) -> torch.Tensor:
        return self.sums

    # -- extras ------------------------------------------------------------------------------------
    def sse(self, ref: torch.Tensor, out: torch.Tensor):
        """out (device double, 1 element) <- sum((ref - recon)^2) over the slab's own rows."""
        lay = self.layout
        own = self.recon_tensor()[lay.row_lo:lay.row_hi]
        refo = ref[lay.row_lo:lay.row_hi]
        _lib.check(_lib.lib().tvdn_sum_square_error(self.ctx, self.code, self.nd, _lib.shape_arr(own.shape),
                                                    refo.data_ptr(), own.data_ptr(), out.data_ptr(),
                                                    _lib.current_stream(self.device)))

    def n_arrays(self) -> int:
        if self.state == "reference":
            return 1 + 2 + 2 * self.nd * (2 if self.fista else 1)
        return 1 + 2 + self.nd * (3 if self.fista else 2)

    def state_bytes(self) -> int:
        return self.n_arrays() * int(np.prod(self.layout.local_shape)) * self.dtype.itemsize


def edge_block(own_rows: int) -> int:
    """Rows swept ahead of the interior at each edge of a slab.  One row would do for the protocol, but a 1-row
    launch pays the fused sweep's prologue and look-ahead rows for a single row of output (measured on a
    66x512x256x256 slab: 24.7 ms per iteration with 1-row edges against 11.5 ms x 2 for the same voxels unsplit);
    a whole 8-row march per side costs nothing extra and still leaves most of the iteration to hide the transfer."""
    e = int(os.environ.get("TVDN_EDGE_ROWS", "8"))
    return max(1, min(e, (own_rows - 1) // 2))


class SlabRunner:
    """Runs iterations on one slab and keeps its halo rows current."""

    def __init__(self, backend, group=None):
        self.be = backend
        self.layout: SlabLayout = backend.layout
        self.group = group
        self.iter = 0
        self.ran = []  # slots of the iterations that actually ran
        self._side = None
        self._p2p = None
        self._stage = None
        self.overlap = True
        if self.layout.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("world > 1 needs an initialised torch.distributed process group")
            self.dist = dist
            # batched P2P must not be the first collective of a group (torch.distributed.batch_isend_irecv note)
            dist.barrier(group=group)

    @property
    def transport(self) -> str:
        """What moves the halo rows: "rccl" (device memory over xGMI), "gloo" (pinned host staging), or None."""
        if self.layout.world == 1:
            return None
        b = self.dist.get_backend(self.group)
        return "rccl" if b == "nccl" else str(b)

    def _device_p2p(self) -> bool:
        """True when the process group can send device memory directly (RCCL); a gloo group moves the
        rows through pinned host buffers instead (used to rehearse multi-rank runs on one GPU)."""
        if self._p2p is None:
            self._p2p = self.dist.get_backend(self.group) != "gloo" or not self.be.recon_tensor().is_cuda
        return self._p2p

    def _exchange_staged(self, r):
        """Same messages as `_ops`, staged through pinned host memory."""
        lay, dist = self.layout, self.dist
        if self._stage is None:
            row = r[lay.row_lo]
            self._stage = [torch.empty(row.shape, dtype=row.dtype, pin_memory=True) for _ in range(6)]
        s_lo, s_hi, r_lo, r_hi, s_wr, r_wr = self._stage
        ops = []
        if lay.left is not None:
            s_lo.copy_(r[lay.row_lo])
        if lay.right is not None:
            s_hi.copy_(r[lay.row_hi - 1])
        if lay.wrap_to is not None:
            s_wr.copy_(r[lay.row_lo])
        torch.cuda.current_stream(r.device).synchronize()
        if lay.left is not None:
            ops.append(dist.P2POp(dist.isend, s_lo, self._peer(lay.left), self.group, tag=1))
        if lay.right is not None:
            ops.append(dist.P2POp(dist.isend, s_hi, self._peer(lay.right), self.group, tag=2))
            ops.append(dist.P2POp(dist.irecv, r_hi, self._peer(lay.right), self.group, tag=1))
        if lay.left is not None:
            ops.append(dist.P2POp(dist.irecv, r_lo, self._peer(lay.left), self.group, tag=2))
        if lay.wrap_to is not None:
            ops.append(dist.P2POp(dist.isend, s_wr, self._peer(lay.wrap_to), self.group, tag=3))
        if lay.wrap_from is not None:
            ops.append(dist.P2POp(dist.irecv, r_wr, self._peer(lay.wrap_from), self.group, tag=3))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if lay.right is not None:
            r[lay.row_hi].copy_(r_hi)
        if lay.left is not None:
            r[lay.row_lo - 1].copy_(r_lo)
        if lay.wrap_from is not None:
            r[lay.row_hi].copy_(r_wr)

    def exchange_halos(self):
        """Send my first/last own recon rows to the neighbours, receive theirs into my halo rows.

        One row = one contiguous block; with RCCL each message rides one
        xGMI link between the two neighbouring GPUs."""
        lay = self.layout
        if lay.world == 1:
            return
        dist = self.dist
        r = self.be.recon_tensor()
        if not self._device_p2p():
            self._exchange_staged(r)
            return
        for w in dist.batch_isend_irecv(self._ops(r)):
            w.wait()

    def _ops(self, r):
        """The four row messages of one exchange.  Order matters where RCCL matches messages per peer in issue
        order and one peer is both neighbours (periodic ring of two): sends go (first row -> left, last row ->
        right), so the receives are posted (from right, from left)."""
        lay, dist = self.layout, self.dist
        ops = []
        if lay.left is not None:
            ops.append(dist.P2POp(dist.isend, r[lay.row_lo], self._peer(lay.left), self.group, tag=1))
        if lay.right is not None:
            ops.append(dist.P2POp(dist.isend, r[lay.row_hi - 1], self._peer(lay.right), self.group, tag=2))
            ops.append(dist.P2POp(dist.irecv, r[lay.row_hi], self._peer(lay.right), self.group, tag=1))
        if lay.left is not None:
            ops.append(dist.P2POp(dist.irecv, r[lay.row_lo - 1], self._peer(lay.left), self.group, tag=2))
        # `wrap_row` layouts: rank 0's first row also goes to the last rank (after the ordinary rows, on both sides, so
        # that with two ranks -- one peer, three messages -- sends and receives still pair up in issue order)
        if lay.wrap_to is not None:
            ops.append(dist.P2POp(dist.isend, r[lay.row_lo], self._peer(lay.wrap_to), self.group, tag=3))
        if lay.wrap_from is not None:
            ops.append(dist.P2POp(dist.irecv, r[lay.row_hi], self._peer(lay.wrap_from), self.group, tag=3))
        return ops

    def step_overlapped(self, tk_ratio, slot: int):
        """One iteration with the halo exchange hidden behind the interior sweep:
        the rows at the two edges are advanced first (a whole march of them per side, `edge_block`, so the two
        extra launches cost no extra look-ahead rows), the transfer of the outermost row of each runs on a side HIP
        stream while the main stream sweeps the interior rows, and the next iteration waits for the transfer."""
        lay, be = self.layout, self.be
        lo, hi = lay.row_lo, lay.row_hi
        if lay.world == 1 or hi - lo < 3 or not getattr(be, "supports_partial_sweeps", False) \
                or not be.recon_tensor().is_cuda or not self._device_p2p():
            be.step(tk_ratio, slot)
            self.exchange_halos()
            return
        main = torch.cuda.current_stream(be.device)
        if self._side is None:
            # HIGH priority = a hardware-queue class the sweeps' stream is not in.  Streams of one priority share a few
            # hardware queues in turn (profiles/r03_queue_alias_probe.jsonl); a side stream that lands on the main stream's
            # queue puts its wait for the exchange IN FRONT of the interior sweep, and the overlap is gone -- one time in
            # four, by the order streams happened to be created in.  (The group's own RCCL stream should be created the
            # same way: ProcessGroupNCCL.Options(is_high_priority_stream=True), as bench.py does.)
            self._side = torch.cuda.Stream(device=be.device, priority=-1)
        main.wait_stream(self._side)                     # the previous exchange has filled my halo rows
        e = edge_block(hi - lo)
        be.step(tk_ratio, slot, rows=(lo, lo + e), accumulate=False)
        be.step(tk_ratio, slot, rows=(hi - e, hi), accumulate=True)
        edge_done = torch.cuda.Event()
        edge_done.record(main)
        r_next = be.recon_next()
        with torch.cuda.stream(self._side):
            self._side.wait_event(edge_done)
            for w in self.dist.batch_isend_irecv(self._ops(r_next)):
                w.wait()                                 # blocks the side stream only
        be.step(tk_ratio, slot, rows=(lo + e, hi - e), accumulate=True)
        be.flip()

    def finish(self):
        """Join the side stream (call before reading recon or the sums)."""
        if self._side is not None:
            torch.cuda.current_stream(self.be.device).wait_stream(self._side)

    def _peer(self, rank_in_group: int) -> int:
        if self.group is None:
            return rank_in_group
        return self.dist.get_global_rank(self.group, rank_in_group)

    def _step(self, tk_ratio, slot):
        if self.overlap:
            self.step_overlapped(tk_ratio, slot)
        else:
            self.be.step(tk_ratio, slot)
            self.exchange_halos()

    def run(self, n_fista: int, n_plain: int, on_iter=None):
        """n_fista FISTA iterations then n_plain unaccelerated ones (hybrid mode of the reference,
        cyTVDN.py:99-108).  `on_iter(slot)` may return True to stop the current phase early."""
        slot = self.iter
        ratios = fista_ratios(n_fista)
        if on_iter is None and self.layout.world == 1 and getattr(self.be, "state", None) == "compact" \
                and hasattr(self.be, "run_many") and os.environ.get("TVDN_LOOP", "run") != "python":
            # nobody watches the iterations: the whole schedule behind one library call
            if n_fista + n_plain:
                self.be.run_many(ratios, n_plain, slot)
            self.ran.extend(range(slot, slot + n_fista + n_plain))
            self.iter += n_fista + n_plain
            return
        for i in range(n_fista):
            self._step(float(ratios[i]), slot)
            self.ran.append(slot)
            slot += 1
            if on_iter is not None and on_iter(slot - 1):
                break
        slot = self.iter + n_fista  # the plain phase starts at its own slot even after an early break
        for _ in range(n_plain):
            self._step(None, slot)
            self.ran.append(slot)
            slot += 1
            if on_iter is not None and on_iter(slot - 1):
                break
        self.iter += n_fista + n_plain
        self.finish()

    def global_sums(self) -> torch.Tensor:
        """[iters,3] f64 sums over ALL slabs (b_norm, sum|delta|, sum|old|)."""
        self.finish()
        s = self.be.sums_tensor().clone()
        if self.layout.world > 1:
            if s.is_cuda and self.dist.get_backend(self.group) == "gloo":
                s = s.cpu()                      # gloo reduces host memory
            self.dist.all_reduce(s, group=self.group)
        return s


class LocalSlabs:
    """Several slabs of one cube advanced in lockstep inside ONE process, halo rows moved by plain
    device copies instead of messages.  Same SlabLayout, same backend calls and the same exchange
    pattern as the multi-process path (it is how the per-slab semantics of the HIP sweep are tested
    on a single GPU, and the building block for staging slabs of a cube that exceeds HBM)."""

    def __init__(self, backends, split_sweeps: bool = False):
        self.bes = list(backends)
        self.world = len(self.bes)
        self.split = split_sweeps  # advance edge rows and interior rows in separate launches
        for r, be in enumerate(self.bes):
            if be.layout.rank != r or be.layout.world != self.world:
                raise ValueError("backends must be given in rank order with world == len(backends)")

    def exchange_halos(self):
        for be in self.bes:
            lay = be.layout
            r = be.recon_tensor()
            if lay.left is not None:
                lb = self.bes[lay.left]
                lb.recon_tensor()[lb.layout.row_hi].copy_(r[lay.row_lo])          # my first row -> left's high halo
            if lay.right is not None:
                rb = self.bes[lay.right]
                rb.recon_tensor()[rb.layout.row_lo - 1].copy_(r[lay.row_hi - 1])  # my last row -> right's low halo
            if lay.wrap_to is not None:
                wb = self.bes[lay.wrap_to]
                wb.recon_tensor()[wb.layout.row_hi].copy_(r[lay.row_lo])          # row 0 -> the last slab's wrap row

    def run(self, n_fista: int, n_plain: int):
        ratios = fista_ratios(n_fista)
        slot = 0
        for i in range(n_fista + n_plain):
            tk = float(ratios[i]) if i < n_fista else None
            for be in self.bes:
                lo, hi = be.layout.row_lo, be.layout.row_hi
                if self.split and hi - lo >= 3:
                    e = edge_block(hi - lo)
                    be.step(tk, slot, rows=(lo, lo + e), accumulate=False)
                    be.step(tk, slot, rows=(hi - e, hi), accumulate=True)
                    be.step(tk, slot, rows=(lo + e, hi - e), accumulate=True)
                    be.flip()
                else:
                    be.step(tk, slot)
            self.exchange_halos()
            slot += 1

    def gather_recon(self) -> torch.Tensor:
        return torch.cat([be.recon_tensor()[be.layout.row_lo:be.layout.row_hi] for be in self.bes], dim=0)

    def global_sums(self) -> torch.Tensor:
        return sum(be.sums_tensor() for be in self.bes)


def hbm_plan(shape, dtype, fista: bool, world: int = 1) -> dict:
    """Bytes of HBM one slab needs in the fused (multi-buffered) engine (cf. check_memory; planner.plan_run decides)."""
    from .planner import state_arrays
    n = int(np.prod(shape)) // max(world, 1)
    item = np.dtype(dtype).itemsize
    arrays = state_arrays(len(shape), fista)   # orig, recon x2, rotating accumulator arrays per axis
    return dict(arrays=arrays, bytes=arrays * n * item, per_array=n * item)


__all__ = ["SlabLayout", "HipBackend", "SlabRunner", "LocalSlabs", "fista_ratios", "hbm_plan", "edge_block"]
