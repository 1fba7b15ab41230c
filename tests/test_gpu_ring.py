"""Row rings of tvdn_iterate_fused (tvdn.h, ring_rows / orig_ring_rows): a sweep over arrays of which only a ring of
row-planes is resident gives the bits of the same sweep over the whole arrays -- every mode, dtype, vector width, both
boundary conditions, edge rows included.  The out-of-core wavefront schedule is built on exactly this."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tv():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import cytvdn_amd
    return cytvdn_amd


def _fill(a, tensors, mode, _lib, ptr_of):
    """Point the state fields of `a` at tensors for `mode`; tensors = dict name -> list per axis / tensor."""
    nd = a.ndim
    a.orig, a.recon_in, a.recon_out = ptr_of(tensors["orig"]), ptr_of(tensors["r_in"]), ptr_of(tensors["r_out"])
    for q in range(nd):
        a.b_in[q] = a.b_out[q] = a.d_in[q] = a.d_out[q] = a.dprev_in[q] = None
        s1, s2, o1, o2 = (tensors[k][q] for k in ("s1", "s2", "o1", "o2"))
        if mode == _lib.ITER_PLAIN:
            a.b_in[q], a.b_out[q] = ptr_of(s1), ptr_of(o1)
        elif mode == _lib.ITER_FISTA:
            a.b_in[q], a.d_in[q], a.b_out[q], a.d_out[q] = ptr_of(s1), ptr_of(s2), ptr_of(o1), ptr_of(o2)
        elif mode == _lib.ITER_FISTA_D:
            a.dprev_in[q], a.d_in[q], a.d_out[q] = ptr_of(s1), ptr_of(s2), ptr_of(o2)
        else:
            a.dprev_in[q], a.d_in[q], a.b_out[q] = ptr_of(s1), ptr_of(s2), ptr_of(o1)


@pytest.mark.parametrize("shape,dtype,bc,R", [
    ((23, 4, 6, 16), np.float32, 2, 4), ((17, 5, 8), np.float64, 2, 3), ((19, 3, 5, 7), np.float32, 2, 5),
    ((16, 4, 6, 8), np.float64, 0, 4), ((21, 6, 12), np.float32, 0, 2), ((13, 3, 6, 16), np.float32, 2, 1), ((11, 4, 8), np.float64, 0, 1),
])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("chained", [False, True], ids=["independent launches", "accumulator handed across the cut"])
def test_ring_sweeps_equal_the_resident_sweep(shape, dtype, bc, R, mode, chained):
    """`chained` (ABI 9, tvdn.h TVDN_SWEEP_*): every launch but the last also stores the axis-0 output state of the row after its
    last, every launch but the first takes the axis-0 accumulator of its first row from there instead of re-reading recon of the
    row before and the input state -- recon of that row is POISONED in the ring to prove it is not read.  Same bits."""
    import torch
    from cytvdn_amd import _lib
    L, ctx = _lib.lib(), _lib.ctx(0)
    nd, N0 = len(shape), shape[0]
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    rng = np.random.default_rng(5 + mode)

    def rand(scale=1.0):
        return torch.from_numpy((rng.standard_normal(shape) * scale).astype(dtype)).cuda()

    full = {"orig": rand(4), "r_in": rand(4), "r_out": torch.zeros(shape, dtype=tdt, device="cuda"),
            "s1": [rand(0.4) for _ in range(nd)], "s2": [rand(0.4) for _ in range(nd)],
            "o1": [torch.zeros(shape, dtype=tdt, device="cuda") for _ in range(nd)],
            "o2": [torch.zeros(shape, dtype=tdt, device="cuda") for _ in range(nd)]}

    for k in ("s1", "s2"):          # Jia-Zhao: the axis-0 accumulator of row 0 is identically zero in any real state
        full[k][0][0].zero_()

    def args():
        a = _lib.IterArgs(dtype=_lib.dtype_code(np.dtype(dtype)), ndim=nd, row_lo=0, row_hi=N0, lo_mode=_lib.EDGE_BC,
                          hi_mode=_lib.EDGE_BC, bc_mode=bc, mode=mode, tk=0.37, tk_prev=0.21, accumulate=1)
        for i, s in enumerate(shape):
            a.shape[i] = s
        for q in range(nd):
            a.clip[q], a.lambda_mu[q] = 0.5 + 0.1 * q, 0.3 - 0.05 * q
        return a

    # the whole cube in one resident sweep
    a = args()
    _fill(a, full, mode, _lib, lambda t: t.data_ptr())
    sums_full = torch.zeros(3, dtype=torch.float64, device="cuda")
    _lib.check(L.tvdn_iterate_fused(ctx, C.byref(a), sums_full.data_ptr(), _lib.current_stream(0)))
    torch.cuda.synchronize()

    # the same cube streamed through rings of R+2 rows (orig: R+1), chunk by chunk.  A periodic wrap needs both ends
    # resident at once, so periodic cubes are swept away from the two faces only (as the wavefront engine does).
    lo, hi = (1, N0 - 1) if bc == 0 else (0, N0)
    cap, ocap = R + 2, R + 1
    plane = shape[1:]

    def ring(n):
        return torch.full((n,) + plane, float("nan"), dtype=tdt, device="cuda")

    rings = {"orig": ring(ocap), "r_in": ring(cap), "r_out": ring(cap), "s1": [ring(cap) for _ in range(nd)],
             "s2": [ring(cap) for _ in range(nd)], "o1": [ring(cap) for _ in range(nd)], "o2": [ring(cap) for _ in range(nd)]}
    got = {"r_out": torch.zeros(shape, dtype=tdt, device="cuda"), "o1": [torch.zeros(shape, dtype=tdt, device="cuda") for _ in range(nd)],
           "o2": [torch.zeros(shape, dtype=tdt, device="cuda") for _ in range(nd)]}
    sums_ring = torch.zeros(3, dtype=torch.float64, device="cuda")
    for c0 in range(lo, hi, R):
        c1 = min(c0 + R, hi)
        for g in range(max(0, c0 - 1), min(N0, c1 + 1)):           # rows the launch reads
            rings["r_in"][g % cap].copy_(full["r_in"][g])
            for q in range(nd):
                rings["s1"][q][g % cap].copy_(full["s1"][q][g])
                rings["s2"][q][g % cap].copy_(full["s2"][q][g])
        for g in range(c0, c1):
            rings["orig"][g % ocap].copy_(full["orig"][g])
        a = args()
        a.sweep_lo, a.sweep_hi = c0, c1
        a.ring_rows, a.orig_ring_rows = cap, ocap
        if chained:
            a.chain = (_lib.SWEEP_CHAIN_LO if c0 > lo else 0) | (_lib.SWEEP_STORE_AHEAD if c1 < hi else 0)
            if c0 > lo:
                rings["r_in"][(c0 - 1) % cap].fill_(float("nan"))      # a chained launch has no use for the row before its first
        if bc == 2:
            a.hi_mode = _lib.EDGE_ZERO            # what a Jia-Zhao cube whose first row is finite has at its top face
        _fill(a, rings, mode, _lib, lambda t: t.data_ptr())
        _lib.check(L.tvdn_iterate_fused(ctx, C.byref(a), sums_ring.data_ptr(), _lib.current_stream(0)))
        for g in range(c0, c1):
            got["r_out"][g].copy_(rings["r_out"][g % cap])
            for q in range(nd):
                got["o1"][q][g].copy_(rings["o1"][q][g % cap])
                got["o2"][q][g].copy_(rings["o2"][q][g % cap])
    torch.cuda.synchronize()

    sl = slice(lo, hi)
    assert torch.equal(got["r_out"][sl].view(torch.uint8), full["r_out"][sl].view(torch.uint8))
    writes1 = mode in (_lib.ITER_PLAIN, _lib.ITER_FISTA, _lib.ITER_FISTA_D_TO_PLAIN)
    writes2 = mode in (_lib.ITER_FISTA, _lib.ITER_FISTA_D)
    for q in range(nd):
        if writes1:
            assert torch.equal(got["o1"][q][sl].view(torch.uint8), full["o1"][q][sl].view(torch.uint8)), ("out1", q)
        if writes2:
            assert torch.equal(got["o2"][q][sl].view(torch.uint8), full["o2"][q][sl].view(torch.uint8)), ("out2", q)
    if bc == 2:     # the sums cover the same rows only then
        np.testing.assert_allclose(sums_ring.cpu().numpy(), sums_full.cpu().numpy(), rtol=1e-12)


@pytest.mark.parametrize("shape,dtype,R", [((23, 4, 6, 16), np.float32, 4), ((17, 5, 8), np.float64, 3), ((19, 3, 5, 7), np.float32, 2)])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("which", ["inputs as arrays", "outputs as arrays", "previous level and recon_out as arrays"])
def test_ring_sizes_per_array_class(shape, dtype, R, mode, which):
    """ABI 8: recon_in, the state of this level (b_in / d_in), of the level before (dprev_in), recon_out and the outputs (b_out /
    d_out) each take a ring size of their own (0 = ring_rows); a ring longer than the cube is an array.  What the streamed engine
    sweeps rows in place with: some classes are whole arrays, the others rings of R + 2 rows, the bits those of the resident sweep."""
    import torch
    from cytvdn_amd import _lib
    L, ctx = _lib.lib(), _lib.ctx(0)
    nd, N0 = len(shape), shape[0]
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    rng = np.random.default_rng(11 + mode)

    def rand(scale=1.0):
        return torch.from_numpy((rng.standard_normal(shape) * scale).astype(dtype)).cuda()

    def zeros():
        return torch.zeros(shape, dtype=tdt, device="cuda")

    full = {"orig": rand(4), "r_in": rand(4), "r_out": zeros(), "s1": [rand(0.4) for _ in range(nd)], "s2": [rand(0.4) for _ in range(nd)],
            "o1": [zeros() for _ in range(nd)], "o2": [zeros() for _ in range(nd)]}
    for k in ("s1", "s2"):
        full[k][0][0].zero_()

    def args():
        a = _lib.IterArgs(dtype=_lib.dtype_code(np.dtype(dtype)), ndim=nd, row_lo=0, row_hi=N0, lo_mode=_lib.EDGE_BC,
                          hi_mode=_lib.EDGE_ZERO, bc_mode=2, mode=mode, tk=0.37, tk_prev=0.21, accumulate=1)
        for i, s_ in enumerate(shape):
            a.shape[i] = s_
        for q in range(nd):
            a.clip[q], a.lambda_mu[q] = 0.5 + 0.1 * q, 0.3 - 0.05 * q
        return a

    a = args()
    a.hi_mode = _lib.EDGE_BC
    _fill(a, full, mode, _lib, lambda t: t.data_ptr())
    _lib.check(L.tvdn_iterate_fused(ctx, C.byref(a), torch.zeros(3, dtype=torch.float64, device="cuda").data_ptr(), _lib.current_stream(0)))
    torch.cuda.synchronize()

    cap = R + 2
    AS_ARRAY = 2 ** 31 - 1
    d_modes = mode in (_lib.ITER_FISTA_D, _lib.ITER_FISTA_D_TO_PLAIN)
    # which tensors of _fill's dictionary belong to which class: s1 is dprev_in in the d modes (the level before), else b_in (this level)
    prev_keys = ["s1"] if d_modes else []
    cur_keys = ["s2"] + ([] if d_modes else ["s1"])
    as_array = {"inputs as arrays": {"r_in", "s1", "s2", "orig"}, "outputs as arrays": {"r_out", "o1", "o2"},
                "previous level and recon_out as arrays": set(prev_keys) | {"r_out"}}[which]

    def ring():
        return torch.full((cap,) + shape[1:], float("nan"), dtype=tdt, device="cuda")

    work = {}
    for k in ("orig", "r_in", "r_out"):
        work[k] = (full[k].clone() if k != "r_out" else zeros()) if k in as_array else ring()
    for k in ("s1", "s2", "o1", "o2"):
        work[k] = [((full[k][q].clone() if k in ("s1", "s2") else zeros()) if k in as_array else ring()) for q in range(nd)]
    got = {"r_out": zeros(), "o1": [zeros() for _ in range(nd)], "o2": [zeros() for _ in range(nd)]}
    sums = torch.zeros(3, dtype=torch.float64, device="cuda")
    for c0 in range(0, N0, R):
        c1 = min(c0 + R, N0)
        for g in range(max(0, c0 - 1), min(N0, c1 + 1)):
            if "r_in" not in as_array:
                work["r_in"][g % cap].copy_(full["r_in"][g])
            for k in ("s1", "s2"):
                if k not in as_array:
                    for q in range(nd):
                        work[k][q][g % cap].copy_(full[k][q][g])
        if "orig" not in as_array:
            for g in range(c0, c1):
                work["orig"][g % cap].copy_(full["orig"][g])
        a = args()
        a.sweep_lo, a.sweep_hi = c0, c1
        a.ring_rows, a.orig_ring_rows = cap, (AS_ARRAY if "orig" in as_array else cap)
        a.recon_in_ring_rows = AS_ARRAY if "r_in" in as_array else 0
        a.cur_ring_rows = AS_ARRAY if all(k in as_array for k in cur_keys) else 0
        a.prev_ring_rows = AS_ARRAY if prev_keys and all(k in as_array for k in prev_keys) else 0
        a.recon_out_ring_rows = AS_ARRAY if "r_out" in as_array else 0
        a.out_ring_rows = AS_ARRAY if "o1" in as_array else 0
        _fill(a, work, mode, _lib, lambda t: t.data_ptr())
        _lib.check(L.tvdn_iterate_fused(ctx, C.byref(a), sums.data_ptr(), _lib.current_stream(0)))
        for g in range(c0, c1):
            got["r_out"][g].copy_(work["r_out"][g] if "r_out" in as_array else work["r_out"][g % cap])
            for k in ("o1", "o2"):
                for q in range(nd):
                    got[k][q][g].copy_(work[k][q][g] if k in as_array else work[k][q][g % cap])
    torch.cuda.synchronize()
    assert torch.equal(got["r_out"].view(torch.uint8), full["r_out"].view(torch.uint8))
    for q in range(nd):
        if mode in (_lib.ITER_PLAIN, _lib.ITER_FISTA, _lib.ITER_FISTA_D_TO_PLAIN):
            assert torch.equal(got["o1"][q].view(torch.uint8), full["o1"][q].view(torch.uint8)), ("out1", q)
        if mode in (_lib.ITER_FISTA, _lib.ITER_FISTA_D):
            assert torch.equal(got["o2"][q].view(torch.uint8), full["o2"][q].view(torch.uint8)), ("out2", q)


def test_ring_argument_checks():
    import torch
    from cytvdn_amd import _lib
    L, ctx = _lib.lib(), _lib.ctx(0)
    shape = (12, 3, 4, 8)
    t = [torch.zeros((6,) + shape[1:], dtype=torch.float32, device="cuda") for _ in range(12)]
    out = torch.zeros(3, dtype=torch.float64, device="cuda")
    a = _lib.IterArgs(dtype=0, ndim=4, row_lo=0, row_hi=12, lo_mode=0, hi_mode=_lib.EDGE_ZERO, bc_mode=2, mode=_lib.ITER_PLAIN)
    for i, s in enumerate(shape):
        a.shape[i] = s
    a.orig, a.recon_in, a.recon_out = t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr()
    for q in range(4):
        a.b_in[q], a.b_out[q] = t[3 + q].data_ptr(), t[7 + q].data_ptr()
        a.clip[q], a.lambda_mu[q] = 1.0, 0.1
    a.sweep_lo, a.sweep_hi = 3, 8

    def err():
        return L.tvdn_last_error().decode()

    a.ring_rows = 6                       # 5 swept rows + the row before + the row after = 7
    assert L.tvdn_iterate_fused(ctx, C.byref(a), out.data_ptr(), None) == -1 and "cannot hold" in err()
    a.ring_rows, a.orig_ring_rows = 0, 6
    assert L.tvdn_iterate_fused(ctx, C.byref(a), out.data_ptr(), None) == -1 and "orig_ring_rows" in err()
    a.ring_rows, a.orig_ring_rows = 6, 3
    a.sweep_lo, a.sweep_hi = 3, 7
    assert L.tvdn_iterate_fused(ctx, C.byref(a), out.data_ptr(), None) == -1 and "orig ring" in err()
    a.orig_ring_rows = 6
    a.hi_mode = _lib.EDGE_WRAP
    assert L.tvdn_iterate_fused(ctx, C.byref(a), out.data_ptr(), None) == -1 and "wrap_recon" in err()
    a.hi_mode = _lib.EDGE_ZERO
    assert L.tvdn_iterate_fused(ctx, C.byref(a), out.data_ptr(), _lib.current_stream(0)) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("shape,dtype,its", [((9, 6, 8, 16), np.float32, [5, 3]), ((11, 10, 12), np.float64, 6)])
def test_ring_instantiation_on_resident_arrays(tv, monkeypatch, shape, dtype, its):
    """TVDN_FORCE_RING runs the ring instantiation with a ring as long as the array: the same rows, so the same
    result as the resident instantiation for a whole denoise."""
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=3, dtype=dt)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    a = fn(x, mu, its, FISTA=True, quiet=True)
    monkeypatch.setenv("TVDN_FORCE_RING", "1")
    b = fn(x, mu, its, FISTA=True, quiet=True)
    for u, v in zip(a, b):
        assert np.asarray(u).tobytes() == np.asarray(v).tobytes()


@pytest.mark.parametrize("shape,dtype,R,first_row_finite", [
    ((13, 3, 4, 8), np.float32, 3, True), ((13, 3, 4, 8), np.float32, 4, False), ((11, 5, 6), np.float64, 2, False),
])
def test_ring_sweeps_against_the_oracle(oracle, shape, dtype, R, first_row_finite):
    """Three FISTA iterations driven through rings of R + 2 rows, launch by launch from this test, against the CPU
    ORACLE directly (not against another HIP run) -- including a non-finite first row, where the top face needs
    TVDN_EDGE_WRAP with the current recon of row 0 handed in as `wrap_recon` (anisotropic.pyx:65-73)."""
    import torch
    from cytvdn_amd import _lib, synth
    L, ctx = _lib.lib(), _lib.ctx(0)
    dt = np.dtype(dtype)
    nd, N0 = len(shape), shape[0]
    tdt = torch.float32 if dt == np.float32 else torch.float64
    x = synth.cube(shape, seed=17, dtype=dt) + dt.type(0.25)
    if not first_row_finite:
        x[(0, 1) + (2,) * (nd - 2)] = np.inf
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    n_it = 3
    ref = oracle.denoise(x, mu, n_it, True)
    if not first_row_finite:
        assert np.isnan(ref["recon"][-1]).any()
    ratios = _lib.fista_ratios(n_it)
    cap, ocap = R + 2, R + 1
    plane = shape[1:]
    # whole-cube arrays are only the "host copies" rows are staged from / to; the sweeps see rings alone
    recon = [torch.from_numpy(x).cuda(), torch.zeros(shape, dtype=tdt, device="cuda")]
    orig = torch.from_numpy(x).cuda()
    S = [[torch.zeros(shape, dtype=tdt, device="cuda") for _ in range(3)] for _ in range(nd)]   # d_k, d_k-1, next

    def ring(n):
        return torch.full((n,) + plane, float("nan"), dtype=tdt, device="cuda")

    r_in, r_out, o_rg = ring(cap), ring(cap), ring(ocap)
    d_in, d_prev, d_out = ([ring(cap) for _ in range(nd)] for _ in range(3))
    tk_prev = 0.0
    i_d, i_prev, i_out = 0, 1, 2
    cur = 0
    for it in range(n_it):
        row0 = recon[cur][0].clone()
        for c0 in range(0, N0, R):
            c1 = min(c0 + R, N0)
            for g in range(max(0, c0 - 1), min(N0, c1 + 1)):
                r_in[g % cap].copy_(recon[cur][g])
                for q in range(nd):
                    d_in[q][g % cap].copy_(S[q][i_d][g])
                    d_prev[q][g % cap].copy_(S[q][i_prev][g])
            for g in range(c0, c1):
                o_rg[g % ocap].copy_(orig[g])
            a = _lib.IterArgs(dtype=_lib.dtype_code(dt), ndim=nd, row_lo=0, row_hi=N0, lo_mode=_lib.EDGE_BC,
                              hi_mode=_lib.EDGE_ZERO if first_row_finite else _lib.EDGE_WRAP, bc_mode=2,
                              mode=_lib.ITER_FISTA_D, tk=float(ratios[it]), tk_prev=tk_prev, accumulate=1)
            for i, s_ in enumerate(shape):
                a.shape[i] = s_
            for q in range(nd):
                a.clip[q] = float((1.0 / lam)[q])
                a.lambda_mu[q] = float((lam / mu).astype(dt)[q])
                a.d_in[q], a.dprev_in[q], a.d_out[q] = d_in[q].data_ptr(), d_prev[q].data_ptr(), d_out[q].data_ptr()
            a.orig, a.recon_in, a.recon_out = o_rg.data_ptr(), r_in.data_ptr(), r_out.data_ptr()
            a.sweep_lo, a.sweep_hi, a.ring_rows, a.orig_ring_rows = c0, c1, cap, ocap
            if not first_row_finite:
                a.wrap_recon = row0.data_ptr()
            sums = torch.zeros(3, dtype=torch.float64, device="cuda")
            _lib.check(L.tvdn_iterate_fused(ctx, C.byref(a), sums.data_ptr(), _lib.current_stream(0)))
            for g in range(c0, c1):
                recon[cur ^ 1][g].copy_(r_out[g % cap])
                for q in range(nd):
                    S[q][i_out][g].copy_(d_out[q][g % cap])
        cur ^= 1
        i_d, i_prev, i_out = i_out, i_d, i_prev
        tk_prev = float(ratios[it])
    torch.cuda.synchronize()
    from golden_util import bits_equal
    assert bits_equal(recon[cur].cpu().numpy(), ref["recon"])
