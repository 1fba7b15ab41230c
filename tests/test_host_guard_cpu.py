"""Host-only pieces of the C ABI, exercised WITHOUT a GPU:
 * tvdn_stream_host_need -- the arithmetic that decides whether a streamed tvdn_run may page-lock its state (the check
   whose absence cost round 2 its GPU boxes): need per mode / dtype / MSE / aliasing, the TVDN_HOST_LIMIT cap, the
   80 % rule, and that a refused call dereferences nothing;
 * tvdn_fista_ratios against the reference's float64 recurrence (cyTVDN/cyTVDN.py:153-156), bit for bit;
 * tvdn_iter_mode / tvdn_roles_bind / tvdn_roles_advance -- the one definition of the role rotation -- against a model
   written out independently here (which array each sweep reads and writes over a hybrid schedule: no array is
   read and written by the same sweep, and every d_k written is the d_k read next)."""
import ctypes as C

import numpy as np
import pytest

from cytvdn_amd import _lib

GiB = 1 << 30


def _args(shape, dtype=0, n_fista=2, n_plain=0, data=1 << 40, recon=1 << 44, mse=False):
    a = _lib.RunArgs(dtype=dtype, ndim=len(shape), bc_mode=2, device=0, n_fista=n_fista, n_plain=n_plain)
    for i, s in enumerate(shape):
        a.shape[i] = s
    a.data, a.recon_out = data, recon           # never dereferenced by tvdn_stream_host_need
    if mse:
        a.reference, a.mse_out = 0x70000000, 0x70000000
    return a


def _need(a):
    need, avail = C.c_int64(), C.c_int64()
    rc = _lib.lib().tvdn_stream_host_need(C.byref(a), C.byref(need), C.byref(avail))
    return rc, need.value, avail.value


@pytest.mark.parametrize("shape,dtype,n_fista,n_plain,mse,cubes", [
    ((8, 4, 4, 8), 0, 3, 0, False, 2 + 4 * 2), ((8, 4, 4, 8), 0, 0, 3, False, 2 + 4), ((8, 4, 8), 1, 2, 1, False, 2 + 3 * 2),
    ((8, 4, 8), 1, 0, 1, True, 3 + 3), ((8, 4, 4, 8), 0, 2, 2, True, 3 + 8),
])
def test_need_counts_the_arrays_a_streamed_run_pins(monkeypatch, shape, dtype, n_fista, n_plain, mse, cubes):
    monkeypatch.setenv("TVDN_HOST_LIMIT", "1G")
    rc, need, avail = _need(_args(shape, dtype, n_fista, n_plain, mse=mse))
    assert rc == 0 and need == cubes * int(np.prod(shape)) * (4 if dtype == 0 else 8) and 0 < avail <= GiB


def test_periodic_runs_keep_old_and_new_state_apart(monkeypatch):
    monkeypatch.setenv("TVDN_HOST_LIMIT", "1G")
    cube = 8 * 4 * 4 * 8 * 4
    a = _args((8, 4, 4, 8))
    a.bc_mode = 0
    assert _need(a)[1] == (2 * (4 * 2 + 1) + 1) * cube          # data + 2 x (recon + 2 state arrays per axis)
    a = _args((8, 4, 8), dtype=1, n_fista=0, n_plain=3)
    a.bc_mode = 0
    assert _need(a)[1] == (2 * (3 + 1) + 1) * 8 * 4 * 8 * 8


def test_aliased_data_and_result_cost_one_more_cube(monkeypatch):
    monkeypatch.setenv("TVDN_HOST_LIMIT", "1G")
    cube = 8 * 4 * 4 * 8 * 4
    assert _need(_args((8, 4, 4, 8), data=0x5000, recon=0x5000))[1] == 11 * cube
    assert _need(_args((8, 4, 4, 8), data=0x5000, recon=0x5000 + cube - 4))[1] == 11 * cube      # partial overlap
    assert _need(_args((8, 4, 4, 8), data=0x5000, recon=0x5000 + cube))[1] == 10 * cube          # adjacent: distinct


def test_the_cap_and_the_80_percent_rule(monkeypatch):
    shape = (256, 256, 128, 128)                       # 4 GiB cube, 40 GiB of pinned state with FISTA
    for cap, ok in (("64G", True), ("51G", True), ("49G", False), ("1G", False), ("512M", False)):
        monkeypatch.setenv("TVDN_HOST_LIMIT", cap)
        rc, need, avail = _need(_args(shape))
        if avail < int(float(cap[:-1]) * (GiB if cap[-1] == "G" else 1 << 20)):
            continue                                   # this machine has less than the cap: nothing to learn from it
        assert need == 10 * 4 * GiB
        assert (rc == 0) == ok, cap
        if not ok:
            assert rc == -2
            msg = _lib.lib().tvdn_last_error().decode()
            assert "host memory" in msg and "exceeds" in msg and str(need) in msg


def test_round_2s_box_killer_is_refused_with_null_arrays(monkeypatch):
    """The call that took two GPU boxes down: an 8 TiB shape behind a 16-byte array.  The arithmetic must refuse it on
    any host, with or without a cap, and must not dereference data / recon_out to do so (NULL here)."""
    big = (4096, 512, 256, 256)
    for cap in (None, "64G", "1G"):
        if cap is None:
            monkeypatch.delenv("TVDN_HOST_LIMIT", raising=False)
        else:
            monkeypatch.setenv("TVDN_HOST_LIMIT", cap)
        rc, need, avail = _need(_args(big, data=0, recon=0))
        assert rc == -2 and need == 10 * 4 * int(np.prod(big)) and avail > 0 and need > avail


def test_argument_errors_and_huge_shapes():
    L = _lib.lib()
    assert L.tvdn_stream_host_need(None, None, None) == -1
    assert L.tvdn_stream_host_need(C.byref(_args((4, 4))), None, None) == -1
    a = _args((4, 0, 4))
    assert L.tvdn_stream_host_need(C.byref(a), None, None) == -1
    a = _args((1 << 40, 1 << 20, 1 << 20, 4))         # 2^84 bytes: beyond int64, must not wrap into "fits"
    rc, need, avail = _need(a)
    assert rc == -2 and need == (1 << 63) - 1


def test_fista_ratios_are_the_reference_recurrence():
    # cyTVDN/cyTVDN.py:153-156:  tk_new = (1 + np.sqrt(1 + 4 * tk ** 2)) / 2 ; ratio = (tk - 1) / tk_new, in float64
    tk, want = 1.0, []
    for _ in range(500):
        tk_new = (1 + np.sqrt(1 + 4 * tk ** 2)) / 2
        want.append((tk - 1) / tk_new)
        tk = tk_new
    got = _lib.fista_ratios(500)
    assert got.dtype == np.float64 and got.tobytes() == np.array(want, np.float64).tobytes()
    assert got[0] == 0.0 and 0.98 < got[-1] < 1.0
    assert _lib.fista_ratios(0).shape == (0,)
    from cytvdn_amd.engine import fista_ratios
    assert fista_ratios(7).tobytes() == got[:7].tobytes()


def test_iter_mode():
    assert _lib.iter_mode(True, True) == _lib.ITER_FISTA_D
    assert _lib.iter_mode(False, True) == _lib.ITER_FISTA_D_TO_PLAIN
    assert _lib.iter_mode(False, False) == _lib.ITER_PLAIN
    with pytest.raises(ValueError, match="cannot follow"):
        _lib.iter_mode(True, False)


@pytest.mark.parametrize("nd,n_fista,n_plain", [(4, 5, 0), (3, 0, 4), (4, 3, 3), (3, 1, 1), (4, 0, 1)])
def test_role_rotation_against_an_independent_model(nd, n_fista, n_plain):
    L = _lib.lib()
    m = _lib.ManyArgs()
    # "addresses": recon 100/101; S[q][k] = 1000 + 10 q + k
    m.recon[0], m.recon[1] = 100, 101
    for q in range(4):
        for k in range(3):
            m.S[q][k] = 1000 + 10 * q + k
    m.cur, m.i_d, m.i_prev, m.i_out, m.i_b, m.i_bout, m.d_form, m.tk_prev = 0, 0, 1, 2, 0, 1, int(n_fista > 0), 0.0
    it = _lib.IterArgs(ndim=nd)
    ratios = _lib.fista_ratios(n_fista)
    # the model: per axis, a dict of what each array HOLDS ("d", k) / ("b", k); recon likewise
    holds = {1000 + 10 * q + k: (("d", 0 if k == 0 else -1) if n_fista else ("b", 0)) for q in range(nd) for k in range(3)}
    holds_r = {100: 0, 101: None}
    last_ratio = 0.0
    for i in range(n_fista + n_plain):
        fista = i < n_fista
        ratio = float(ratios[i]) if fista else 0.0
        assert L.tvdn_roles_bind(C.byref(m), int(fista), ratio, C.byref(it)) == 0
        assert holds_r[it.recon_in] == i and it.recon_out != it.recon_in
        assert it.tk == ratio and it.tk_prev == last_ratio
        for q in range(4):
            if q >= nd:
                assert not any((it.b_in[q], it.b_out[q], it.d_in[q], it.d_out[q], it.dprev_in[q]))
                continue
            if fista:
                assert it.mode == _lib.ITER_FISTA_D and not it.b_in[q] and not it.b_out[q]
                assert holds[it.d_in[q]] == ("d", i) and holds[it.dprev_in[q]] == ("d", i - 1)
                assert it.d_out[q] not in (it.d_in[q], it.dprev_in[q])
                holds[it.d_out[q]] = ("d", i + 1)
            elif it.mode == _lib.ITER_FISTA_D_TO_PLAIN:
                assert i == n_fista and n_fista > 0 and not it.b_in[q] and not it.d_out[q]
                assert holds[it.d_in[q]] == ("d", i) and holds[it.dprev_in[q]] == ("d", i - 1)
                assert it.b_out[q] not in (it.d_in[q], it.dprev_in[q])
                holds[it.b_out[q]] = ("b", i + 1)
            else:
                assert it.mode == _lib.ITER_PLAIN and not it.d_in[q] and not it.d_out[q] and not it.dprev_in[q]
                assert holds[it.b_in[q]] == ("b", i) and it.b_out[q] != it.b_in[q]
                holds[it.b_out[q]] = ("b", i + 1)
        holds_r[it.recon_out] = i + 1
        assert L.tvdn_roles_advance(C.byref(m), int(fista), ratio) == 0
        if fista:
            last_ratio = ratio
    assert m.cur == (n_fista + n_plain) % 2 and bool(m.d_form) == (n_plain == 0 and n_fista > 0)
    if n_plain:
        assert L.tvdn_roles_bind(C.byref(m), 1, 0.5, C.byref(it)) == -1      # FISTA after unaccelerated: refused


def test_the_pipelining_plan_switches_itself_on_for_large_cubes_only(monkeypatch):
    """tvdn_pipeline_plan (csrc/tvdn_run.hip): the shape tvdn_run gives a resident run's overlapped transfers."""
    from cytvdn_amd import _lib
    monkeypatch.delenv("TVDN_PIPELINE", raising=False)
    assert _lib.pipeline_plan(256, 50, 4 << 30) == (32, 8, 8)
    assert _lib.pipeline_plan(256, 6, 4 << 30) == (32, 3, 3)
    assert _lib.pipeline_plan(256, 3, 4 << 30) is None          # too few iterations to hide anything under
    assert _lib.pipeline_plan(16, 50, 4 << 30) is None
    assert _lib.pipeline_plan(256, 50, 64 << 20) is None        # small cubes: two transfers of milliseconds
    assert _lib.pipeline_plan(40, 50, 4 << 30) == (8, 8, 8)     # never chunks under eight rows
    monkeypatch.setenv("TVDN_PIPELINE", "0")
    assert _lib.pipeline_plan(256, 50, 4 << 30) is None
    monkeypatch.setenv("TVDN_PIPELINE", "5,6,6")
    assert _lib.pipeline_plan(40, 9, 1 << 20) == (5, 6, 3)      # forced: clipped to the iterations there are
    monkeypatch.setenv("TVDN_PIPELINE", "1")
    assert _lib.pipeline_plan(256, 50, 4 << 30) == (32, 8, 8)


def test_workspace_bytes_is_the_layout_of_the_resident_state():
    """tvdn_run_workspace_bytes: arrays of the cube's size rounded up to 256 bytes and staggered by 4 KiB, 3 + ndim x (3 with
    FISTA iterations, else 2) of them -- host arithmetic, no GPU."""
    import ctypes as C
    from cytvdn_amd import _lib
    for shape, dtype, n_f, want_arrays in (((256, 256, 128, 128), 0, 50, 15), ((256, 256, 128, 128), 1, 0, 11),
                                           ((128, 128, 512), 0, 200, 12), ((5, 3, 7), 1, 0, 9)):
        a = _lib.RunArgs(dtype=dtype, ndim=len(shape), n_fista=n_f, n_plain=3)
        for i, s in enumerate(shape):
            a.shape[i] = s
        need = C.c_int64()
        _lib.check(_lib.lib().tvdn_run_workspace_bytes(C.byref(a), C.byref(need)))
        cube = int(np.prod(shape)) * (4 if dtype == 0 else 8)
        assert need.value == want_arrays * (-(-cube // 256) * 256 + 4096)
    a = _lib.RunArgs(dtype=0, ndim=5)
    assert _lib.lib().tvdn_run_workspace_bytes(C.byref(a), C.byref(need)) == -1


def test_which_rows_of_a_slab_stay_resident_and_where_the_others_sit():
    """tvdn_slab_row_map: the library's own map of a slab's rows (csrc/tvdn_stream.hip RowMap::slab_window), no GPU needed.  For
    every cut, depth and resident count: exactly that many rows are resident, none of them among the `depth` rows at a face
    shared with a neighbour (those are what the exchange hook sends), the host rows fill the packed local arrays without a gap
    behind the `depth` halo rows, and the resident rows are spread evenly over the interior."""
    import ctypes as C
    import itertools
    from cytvdn_amd import _lib
    L = _lib.lib()
    for n0, world, depth, bc in itertools.product((9, 20, 64, 128), (2, 3, 8), (1, 3, 16), (0, 2)):
        cuts = [r * n0 // world for r in range(world + 1)]
        for rank in range(world):
            g0, g1 = cuts[rank], cuts[rank + 1]
            own = g1 - g0
            if own < depth or own < 1:
                continue
            shared_lo, shared_hi = bc == 0 or rank > 0, bc == 0 or rank < world - 1
            interior = max(0, own - depth * (int(shared_lo) + int(shared_hi)))
            io = _lib.SlabIO(global_rows=n0, row0=g0, rank=rank, world=world)
            a = _lib.RunArgs(dtype=0, ndim=4, bc_mode=bc)
            for i, v in enumerate((own, 2, 3, 4)):
                a.shape[i] = v
            a.slab = C.pointer(io)
            slots = (C.c_int64 * own)()
            for res in sorted({0, 1, interior // 2, interior}):
                if res > interior:
                    continue
                assert L.tvdn_slab_row_map(C.byref(a), depth, res, slots) == 0, L.tvdn_last_error()
                s = list(slots)
                assert s.count(-1) == res
                if shared_lo:
                    assert all(v >= 0 for v in s[:depth])
                if shared_hi:
                    assert all(v >= 0 for v in s[own - depth:])
                assert [v for v in s if v >= 0] == list(range(depth, depth + own - res))       # packed, in order, behind the halo
                if res > 1:                                                                     # evenly spread over the interior
                    idx = [i for i, v in enumerate(s) if v < 0]
                    gaps = [b - a_ for a_, b in zip(idx, idx[1:])]
                    assert max(gaps) - min(gaps) <= 1
            assert L.tvdn_slab_row_map(C.byref(a), depth, interior + 1, slots) != 0                # more than the interior holds


def _stream_plan(shape, hbm, n_f=200, n_p=0, resident=-1, stop=False, bc=2, dtype=0):
    import ctypes as C
    from cytvdn_amd import _lib
    a = _lib.RunArgs(dtype=dtype, ndim=len(shape), bc_mode=bc, n_fista=n_f, n_plain=n_p, use_stop=int(stop), stop=0.01,
                     stream_rows=-1, stream_k=-1, stream_resident=resident)
    for i, v in enumerate(shape):
        a.shape[i] = v
    po = _lib.StreamPlanOut()
    rc = _lib.lib().tvdn_stream_plan(C.byref(a), int(hbm), C.byref(po))
    return rc, po


def test_the_streamed_plan_as_arithmetic():
    """tvdn_stream_plan with the free HBM given is pure arithmetic (csrc/tvdn_stream.hip choose_stream_shape), checked here on
    BASELINE's planes: what it plans fits 85 % of the HBM it was given; a rank slab of config 5 keeps nothing and takes
    one-row chunks at the deepest k (a level costs R + 2 rows per array); half of it keeps most rows and takes two-row
    chunks; a stopping rule means one level per pass; more HBM never plans a shallower run; a periodic cube keeps nothing."""
    gib = 2 ** 30
    plane = 1024 * 256 * 256 * 4
    for hbm in (200 * gib, 268 * gib, 288 * gib):
        rc, full = _stream_plan((128, 1024, 256, 256), hbm)
        assert rc == 0 and full.hbm_bytes <= 0.85 * hbm
        assert (full.rows, full.resident_rows) == (1, 0) and full.k >= 25          # PCIe-bound: depth is speed (200 iterations: 29 = 7 passes, 34 = 6, 40 = 5)
        assert full.host_bytes == 10 * 128 * plane                                   # data term, recon, 8 accumulator arrays
        rc, half = _stream_plan((64, 1024, 256, 256), hbm)
        assert rc == 0 and half.hbm_bytes <= 0.85 * hbm
        if hbm >= 268 * gib:
            assert half.rows >= 2 and half.resident_rows == 64 and 3 <= half.k <= 16    # every row fits beside the rings of the levels in between (the lean layout): tall chunks, shallow rings
            assert half.host_bytes == 10 * (64 - half.resident_rows) * plane
        rc, none = _stream_plan((64, 1024, 256, 256), hbm, resident=0)
        assert rc == 0 and none.resident_rows == 0 and none.rows == 1 and none.k >= half.k and (none.k > half.k or half.resident_rows == 0)
        rc, st = _stream_plan((64, 1024, 256, 256), hbm, stop=True)
        assert rc == 0 and st.k == 1
        rc, per = _stream_plan((64, 1024, 256, 256), hbm, bc=0)
        assert rc == 0 and per.resident_rows == 0 and per.host_bytes == (2 * 9 + 1) * 64 * plane     # old and new state apart
    ks = [_stream_plan((128, 1024, 256, 256), g * gib)[1].k for g in (80, 120, 160, 200, 240, 280)]
    assert ks == sorted(ks) and ks[0] >= 1
    rc, _ = _stream_plan((128, 1024, 256, 256), 8 * gib)                             # not even one level of one-row... two-row chunks
    assert rc != 0
    # f64 planes are twice as heavy: shallower at the same HBM
    assert _stream_plan((128, 1024, 256, 256), 268 * gib, dtype=1)[1].k < _stream_plan((128, 1024, 256, 256), 268 * gib)[1].k
