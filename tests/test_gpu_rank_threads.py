"""The per-process slab path of the streamed engine (tvdn_run with a tvdn_slab_io, csrc/tvdn_stream.hip run_streamed_rank) driven
WITHOUT processes: every "rank" is a thread of this process that calls tvdn_run with its own slab and hooks written in Python
(the K-row swap through shared copies and a barrier, the all-reduce by adding up in rank order, the row-0 broadcast through a
mailbox).  Same library path as cytvdn_amd.distributed.denoise_slabs(staged=...), which tests/test_gpu_two_ranks.py runs over
gloo in real processes -- here cheap enough for property tests over cuts, depths, boundary conditions, stopping rule and
non-finite first rows.  The yardstick is the CPU oracle, bit for bit."""
import ctypes as C
import threading

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from golden_util import bits_equal

pytestmark = pytest.mark.gpu


def _view(ptr, n, row_bytes):
    raw = np.ctypeslib.as_array(C.cast(C.c_void_p(ptr), C.POINTER(C.c_uint8)), shape=(int(n) * int(row_bytes),))
    return raw.reshape(int(n), int(row_bytes))


def run_ranks_in_threads(x, mu, n_f, n_p, world, rows, k, bc=2, stop=None, cuts=None, resident=-1):
    """Returns (recon of the whole cube, sums added over the ranks, iterations run, rows each rank kept resident in HBM).
    `resident`: tvdn_run_args.stream_resident of every rank (-1: as many interior rows as fit, 0: none, n: at most n)."""
    from cytvdn_amd import _lib
    N0, nd, dt = x.shape[0], x.ndim, x.dtype
    cuts = cuts or [r * N0 // world for r in range(world + 1)]
    periodic = bc == 0
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    n = n_f + n_p
    bad = bool(bc == 2 and not np.isfinite(x[0]).all())
    bar = threading.Barrier(world)
    lock = threading.Condition()
    post, sums_box, mail = {}, [None] * world, {"sent": 0, "planes": None, "taken": [0] * world}
    errors, out = [], [None] * world

    def rank_main(rank):
        left = (rank - 1) % world if (rank > 0 or periodic) else None
        right = (rank + 1) % world if (rank < world - 1 or periodic) else None

        def exchange(_u, n_arr, arrays, rows_per, lo, hi, depth, rb):
            try:
                vs = [_view(arrays[i], rows_per, rb) for i in range(n_arr)]
                post[(rank, "hi")] = [v[hi - depth:hi].copy() for v in vs]
                post[(rank, "lo")] = [v[lo:lo + depth].copy() for v in vs]
                bar.wait()
                if left is not None:
                    for v, src in zip(vs, post[(left, "hi")]):
                        v[lo - depth:lo] = src
                if right is not None:
                    for v, src in zip(vs, post[(right, "lo")]):
                        v[hi:hi + depth] = src
                bar.wait()
                return 0
            except Exception as e:                      # noqa: BLE001 -- never through the C frames
                errors.append(e)
                bar.abort()
                return 1

        def allreduce(_u, s3):
            try:
                sums_box[rank] = [s3[0], s3[1], s3[2]]
                bar.wait()
                tot = [sum(b[j] for b in sums_box) for j in range(3)]
                bar.wait()
                for j in range(3):
                    s3[j] = tot[j]
                return 0
            except Exception as e:                      # noqa: BLE001
                errors.append(e)
                bar.abort()
                return 1

        def relay(_u, send, planes, n_planes, rb):
            v = _view(planes, n_planes, rb)
            with lock:
                if send:
                    mail["planes"] = v.copy()
                    mail["sent"] += 1
                    lock.notify_all()
                else:
                    if not lock.wait_for(lambda: mail["sent"] > mail["taken"][rank] or errors, timeout=60):
                        errors.append(TimeoutError("row-0 planes never arrived"))
                        return 1
                    v[...] = mail["planes"]
                    mail["taken"][rank] += 1
            return 0

        try:
            g0, g1 = cuts[rank], cuts[rank + 1]
            own = np.ascontiguousarray(x[g0:g1])
            io = _lib.SlabIO(global_rows=N0, row0=g0, rank=rank, world=world, first_row_nonfinite=int(bad))
            cbs = (_lib.SLAB_EXCHANGE(exchange), _lib.SLAB_ALLREDUCE(allreduce), _lib.SLAB_RELAY(relay))
            io.exchange, io.allreduce, io.relay_row0 = cbs
            a = _lib.RunArgs(dtype=_lib.dtype_code(dt), ndim=nd, bc_mode=bc, device=0, n_fista=n_f, n_plain=n_p,
                             use_stop=int(stop is not None), stop=float(stop or 0.0), stream_rows=rows, stream_k=k,
                             stream_resident=resident)
            for i, v in enumerate(own.shape):
                a.shape[i] = int(v)
            for q in range(nd):
                a.clip[q] = float((1.0 / lam)[q])
                a.lambda_mu[q] = float((lam / mu).astype(dt)[q])
            recon, sums, ran, st = np.empty_like(own), np.zeros((max(n, 1), 3)), C.c_int32(0), _lib.RunStats()
            a.data, a.recon_out, a.sums_out, a.stats = own.ctypes.data, recon.ctypes.data, sums.ctypes.data, C.addressof(st)
            a.iters_run = C.addressof(ran)
            a.slab = C.pointer(io)
            rc = _lib.lib().tvdn_run(C.byref(a))
            if rc:
                errors.append(RuntimeError(f"rank {rank}: {_lib.lib().tvdn_last_error().decode()}"))
                bar.abort()
            out[rank] = (recon, sums[:n], ran.value, int(st.resident_rows))
        except Exception as e:                          # noqa: BLE001
            errors.append(e)
            bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    real = [e for e in errors if not isinstance(e, threading.BrokenBarrierError)]
    if real or errors:
        raise (real or errors)[0]
    return np.concatenate([o[0] for o in out], axis=0), sum(o[1] for o in out), out[0][2], [o[3] for o in out]


def _cube(shape, dt, seed, bad):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(shape) * 2 + rng.poisson(3.0, shape)).astype(dt)
    if bad:
        x[(0,) + tuple(int(rng.integers(s)) for s in shape[1:])] = np.inf
    return x


@pytest.mark.parametrize("world,shape,dtype,n_f,n_p,rows,k,bc,stop,bad,resident", [
    (2, (20, 3, 4, 8), np.float32, 9, 0, 4, 3, 2, None, False, -1),   # every interior row resident: rank 0 keeps 10 - 3, rank 1 too
    (2, (20, 3, 4, 8), np.float32, 9, 0, 4, 3, 2, None, False, 0),    # none
    (2, (20, 3, 4, 8), np.float32, 9, 0, 4, 3, 2, None, False, 4),    # some, spread over the interior
    (3, (19, 6, 16), np.float64, 5, 4, 3, 4, 2, None, False, -1),     # hybrid, uneven slabs (the middle one has 7 - 8 < 0 interior rows)
    (3, (30, 6, 16), np.float64, 5, 4, 3, 4, 2, None, False, 3),
    (3, (9, 3, 4, 8), np.float32, 7, 0, 2, 3, 2, None, True, -1),     # the middle rank's halo reaches the top face: it needs row 0 too
    (2, (16, 3, 4, 8), np.float32, 6, 0, 2, 2, 2, None, True, -1),    # non-finite first row AND row 0 itself resident on rank 0
    (3, (18, 3, 4, 8), np.float32, 6, 0, 2, 4, 0, None, False, -1),   # periodic: a ring of slabs (keeps none)
    (2, (12, 5, 12), np.float64, 0, 8, 5, 2, 0, 0.02, False, -1),     # periodic with a stopping rule
    (4, (13, 2, 3, 4), np.float32, 8, 0, 1, 3, 2, 0.02, True, -1),    # stopping rule + non-finite first row, one-row chunks
    (2, (22, 2, 3, 4), np.float32, 8, 0, 3, 1, 2, 0.02, False, 5),    # stopping rule (one level per pass) with resident rows
])
def test_ranks_as_threads(oracle, world, shape, dtype, n_f, n_p, rows, k, bc, stop, bad, resident):
    dt = np.dtype(dtype)
    nd = len(shape)
    x = _cube(shape, dt, 7, bad)
    mu = np.array([1.0, 0.7, 0.5, 1.3][:nd], dt)
    k = min(k, shape[0] // world)                      # what distributed.denoise_slabs does: a pass reads k rows of the neighbour's
    recon, sums, ran, kept = run_ranks_in_threads(x, mu, n_f, n_p, world, rows, k, bc, stop, resident=resident)
    its = [n_f, n_p] if (n_f and n_p) else (n_f or n_p)
    ref = oracle.denoise(x, mu, its, n_f > 0, BC_mode=bc, stopping_relative_change=stop)
    assert bits_equal(recon, ref["recon"])
    # what each rank may keep: its own rows but the k at every face it shares with a neighbour (periodic runs keep none)
    cuts = [r * shape[0] // world for r in range(world + 1)]
    n_total = n_f + n_p
    kk = 1 if stop is not None else -(-n_total // -(-n_total // min(k, n_total)))    # passes of (almost) equal depth: the deepest one
    interior = [max(0, cuts[r + 1] - cuts[r] - (kk if r > 0 else 0) - (kk if r < world - 1 else 0)) for r in range(world)]
    want = [0] * world if bc == 0 or resident == 0 else [i if resident < 0 else min(i, resident) for i in interior]
    assert kept == want, (kept, want)
    if stop is None:
        assert ran == n_f + n_p
        if not bad:
            np.testing.assert_allclose(sums[:, 0], ref["b_norm64"][:ran], rtol=1e-9)
            np.testing.assert_allclose(sums[:, 1], ref["delta64"][:ran], rtol=1e-9)
    else:
        assert ran == ref["iters_done"]


@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(world=st.integers(2, 4), per=st.integers(1, 7), extra=st.integers(0, 3),
       plane=st.one_of(st.tuples(st.integers(1, 3), st.integers(2, 4), st.sampled_from([4, 8, 12])), st.tuples(st.integers(2, 5), st.sampled_from([4, 7, 16]))),
       f64=st.booleans(), bc=st.sampled_from([0, 2]), n_f=st.integers(0, 6), n_p=st.integers(0, 4), seed=st.integers(0, 2 ** 31 - 1),
       chunk=st.integers(1, 5), k=st.integers(1, 7), bad=st.booleans(), stop=st.booleans(), resident=st.sampled_from([-1, -1, 0, 1, 2, 5]))
def test_ranks_as_threads_random(oracle, world, per, extra, plane, f64, bc, n_f, n_p, seed, chunk, k, bad, stop, resident):
    if n_f + n_p == 0:
        n_f = 2
    rows = world * per + extra
    if bc == 0 and rows < 3:
        rows = 3
    shape = (rows,) + tuple(plane)
    dt = np.dtype(np.float64 if f64 else np.float32)
    nd = len(shape)
    x = _cube(shape, dt, seed, bad and bc == 2)
    mu = np.array([1.0, 0.7, 0.5, 1.3][:nd], dt)
    stop_v = 0.02 if stop else None
    k = max(1, min(k, rows // world))
    recon, sums, ran, kept = run_ranks_in_threads(x, mu, n_f, n_p, world, chunk, k, bc, stop_v, resident=resident)
    its = [n_f, n_p] if (n_f and n_p) else (n_f or n_p)
    ref = oracle.denoise(x, mu, its, n_f > 0, BC_mode=bc, stopping_relative_change=stop_v)
    assert bits_equal(recon, ref["recon"])
    assert ran == (ref["iters_done"] if stop_v is not None else n_f + n_p)
    assert all(kk >= 0 for kk in kept) and (bc == 2 or not any(kept))
