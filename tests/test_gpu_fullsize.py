"""Parity at BASELINE.json's full sizes (config 2/3 shape 256x256x128x128, config 1 shape), through
size-independent properties:

 * locality: after k iterations a voxel depends on the input within L1 distance 2k only, so a window of the
   full-size result must equal, bit for bit, the oracle run on that window of the input enlarged by a 2k halo
   (windows at cube corners use the true boundary on the sides that coincide with it);
 * determinism: two runs give identical bits;
 * representation independence: compact (d-rotation) and reference (b, d) state give identical bits;
 * partition independence: 2 logical slabs == 1 slab at full size;
 * scalar conservation: the per-slab sums add up to the single-slab sums (1e-12).
"""
import hashlib

import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu

SHAPE = (256, 256, 128, 128)
K = 3  # iterations


def _sha(t):
    return hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()


@pytest.fixture(scope="module")
def full_run():
    """Config-2 input synthesised in HBM, K FISTA iterations in both state representations."""
    import torch
    from cytvdn_amd import _lib, synth
    from cytvdn_amd.engine import HipBackend, SlabLayout, SlabRunner
    assert torch.cuda.is_available()
    dt = np.dtype(np.float32)
    mu = np.array([1.0, 1.0, 0.5, 0.5], dt)
    lam = mu / dt.type(32.0)
    out = {}
    for state in ("compact", "reference"):
        lay = SlabLayout(SHAPE, 0, 1, 2)
        be = HipBackend(lay, dt, True, device=0, max_iters=K, state=state)
        be.set_params(1.0 / lam, (lam / mu).astype(dt))
        _lib.check(_lib.lib().tvdn_synth_fill(be.code, 4, _lib.shape_arr(SHAPE), synth.SEED_4D, 0, SHAPE[0],
                                              be.orig.data_ptr(), _lib.current_stream(0)))
        be.recon[be.cur].copy_(be.orig)
        SlabRunner(be).run(K, 0)
        out[state] = dict(recon=be.recon_tensor().clone(), sums=be.sums.cpu().numpy().copy())
        if state == "compact":
            out["orig"] = be.orig.clone()
        del be
        torch.cuda.empty_cache()
    out["mu"], out["lam"] = mu, lam
    return out


def test_state_representations_agree_at_full_size(full_run):
    assert _sha(full_run["compact"]["recon"]) == _sha(full_run["reference"]["recon"])
    np.testing.assert_allclose(full_run["compact"]["sums"], full_run["reference"]["sums"], rtol=1e-12)


WINDOWS = [
    # (start index per axis, extent per axis): corners, faces, interior
    ((0, 0, 0, 0), (10, 9, 12, 16)),
    ((246, 247, 116, 112), (10, 9, 12, 16)),
    ((100, 0, 60, 112), (8, 8, 8, 16)),
    ((0, 200, 0, 40), (7, 9, 10, 12)),
    ((131, 77, 59, 64), (9, 8, 11, 12)),
]


@pytest.mark.parametrize("start,ext", WINDOWS, ids=lambda v: "-".join(map(str, v)))
def test_locality_window_matches_oracle(oracle, full_run, start, ext):
    halo = 2 * K
    lo = [max(0, s - halo) for s in start]
    hi = [min(n, s + e + halo) for s, e, n in zip(start, ext, SHAPE)]
    sl = tuple(slice(a, b) for a, b in zip(lo, hi))
    x = full_run["orig"][sl].cpu().numpy().copy()
    ref = oracle.denoise(x, full_run["mu"], K, True)["recon"]
    # inside the enlarged window, only voxels at least `halo` away from an ARTIFICIAL face are exact;
    # faces that coincide with the true cube boundary are exact as they are
    inner = tuple(slice(s - a, s - a + e) for s, a, e in zip(start, lo, ext))
    got = full_run["compact"]["recon"][tuple(slice(s, s + e) for s, e in zip(start, ext))].cpu().numpy()
    assert bits_equal(got, ref[inner])


def test_determinism_and_two_slabs_at_full_size(full_run):
    import torch
    from cytvdn_amd import _lib, synth
    from cytvdn_amd.engine import HipBackend, LocalSlabs, SlabLayout
    dt = np.dtype(np.float32)
    mu, lam = full_run["mu"], full_run["lam"]
    bes = []
    for r in range(2):
        lay = SlabLayout(SHAPE, r, 2, 2)
        be = HipBackend(lay, dt, True, device=0, max_iters=K)
        be.set_params(1.0 / lam, (lam / mu).astype(dt))
        _lib.check(_lib.lib().tvdn_synth_fill(be.code, 4, _lib.shape_arr(SHAPE), synth.SEED_4D, lay.g0 - lay.halo_lo,
                                              lay.local_shape[0], be.orig.data_ptr(), _lib.current_stream(0)))
        be.recon[be.cur].copy_(be.orig)
        bes.append(be)
    grp = LocalSlabs(bes, split_sweeps=True)
    grp.run(K, 0)
    recon = grp.gather_recon()
    assert _sha(recon) == _sha(full_run["compact"]["recon"])       # same bits as the single slab, run earlier
    np.testing.assert_allclose(grp.global_sums().cpu().numpy(), full_run["compact"]["sums"], rtol=1e-12)
    del grp, bes, recon
    torch.cuda.empty_cache()


def test_config3_f64_plain_locality(oracle):
    """Config 3 (float64, unaccelerated) at full size: one window against the oracle."""
    import torch
    from cytvdn_amd import _lib, synth
    from cytvdn_amd.engine import HipBackend, SlabLayout, SlabRunner
    dt = np.dtype(np.float64)
    mu = np.array([1.0, 1.0, 0.5, 0.5], dt)
    lam = mu / 32.0
    be = HipBackend(SlabLayout(SHAPE, 0, 1, 2), dt, False, device=0, max_iters=K)
    be.set_params(1.0 / lam, lam / mu)
    _lib.check(_lib.lib().tvdn_synth_fill(be.code, 4, _lib.shape_arr(SHAPE), synth.SEED_4D, 0, SHAPE[0],
                                          be.orig.data_ptr(), _lib.current_stream(0)))
    be.recon[be.cur].copy_(be.orig)
    SlabRunner(be).run(0, K)
    start, ext = (250, 3, 120, 0), (6, 8, 8, 16)
    halo = 2 * K
    lo = [max(0, s - halo) for s in start]
    hi = [min(n, s + e + halo) for s, e, n in zip(start, ext, SHAPE)]
    x = be.orig[tuple(slice(a, b) for a, b in zip(lo, hi))].cpu().numpy().copy()
    ref = oracle.denoise(x, mu, K, False)["recon"]
    inner = tuple(slice(s - a, s - a + e) for s, a, e in zip(start, lo, ext))
    got = be.recon_tensor()[tuple(slice(s, s + e) for s, e in zip(start, ext))].cpu().numpy()
    assert bits_equal(got, ref[inner])
    del be
    torch.cuda.empty_cache()


@pytest.mark.parametrize("shape,fista", [
    ((520, 256, 128, 128), True),      # 2.18e9 elements per array: past the signed 32-bit range (8.1 GiB x 15 arrays)
    ((1056, 256, 128, 128), False),    # 4.43e9 elements per array: past the unsigned 32-bit range (16.5 GiB x 11 arrays)
], ids=["past-2^31-FISTA", "past-2^32-plain"])
def test_large_offsets_locality(oracle, shape, fista):
    """Maximum-size regime: element offsets beyond 32 bits.  Windows at the far end of the cube against the oracle."""
    import torch
    from cytvdn_amd import _lib, synth
    from cytvdn_amd.engine import HipBackend, SlabLayout, SlabRunner
    dt = np.dtype(np.float32)
    k = 2
    mu = np.array([1.0, 1.0, 0.5, 0.5], dt)
    lam = mu / dt.type(32.0)
    be = HipBackend(SlabLayout(shape, 0, 1, 2), dt, fista, device=0, max_iters=k)
    be.set_params(1.0 / lam, (lam / mu).astype(dt))
    _lib.check(_lib.lib().tvdn_synth_fill(be.code, 4, _lib.shape_arr(shape), synth.SEED_4D, 0, shape[0],
                                          be.orig.data_ptr(), _lib.current_stream(0)))
    be.recon[be.cur].copy_(be.orig)
    SlabRunner(be).run(k if fista else 0, 0 if fista else k)
    halo = 2 * k
    for start, ext in (((shape[0] - 9, 250, 119, 96), (9, 6, 9, 16)), ((shape[0] // 2 + 3, 0, 0, 0), (6, 7, 8, 12))):
        lo = [max(0, s - halo) for s in start]
        hi = [min(n, s + e + halo) for s, e, n in zip(start, ext, shape)]
        x = be.orig[tuple(slice(a, b) for a, b in zip(lo, hi))].cpu().numpy().copy()
        ref = oracle.denoise(x, mu, k, fista)["recon"]
        inner = tuple(slice(s - a, s - a + e) for s, a, e in zip(start, lo, ext))
        got = be.recon_tensor()[tuple(slice(s, s + e) for s, e in zip(start, ext))].cpu().numpy()
        assert bits_equal(got, ref[inner])
    # the sums of a cube this size are still finite and positive
    s = be.sums.cpu().numpy()
    assert np.all(np.isfinite(s)) and np.all(s[:, 2] > 0)
    del be
    torch.cuda.empty_cache()


@pytest.mark.parametrize("plane,own,thin,k", [
    ((512, 256, 256), 64, 3, 3),     # BASELINE configs[3]: one GPU's slab, 66-row local block of 128 MiB rows (124 GiB of state)
    ((1024, 256, 256), 40, 2, 2),    # configs[4] planes (256 MiB rows): 2^16 tiles x 5 marches = 327 680 workgroups
], ids=["config4-slab-66x512x256x256", "config5-plane-42x1024x256x256"])
def test_multi_gpu_slab_shape_on_one_gpu(oracle, plane, own, thin, k):
    """The slab one GPU holds in the 8-GPU configurations, swept exactly as `step_overlapped` sweeps it (halo rows on
    both sides, edge rows first, then the interior), with two thin neighbour slabs on the same GPU supplying true
    halo rows.  Windows of the result (slab edges included) against the oracle; sums against the thin-slab-free
    total are covered by the smaller slab tests."""
    import gc
    import torch
    from cytvdn_amd import _lib, synth
    from cytvdn_amd.engine import HipBackend, LocalSlabs, SlabLayout
    gc.collect()
    torch.cuda.empty_cache()                    # 140-180 GiB in one piece: start from an empty allocator
    dt = np.dtype(np.float32)
    shape = (thin + own + thin,) + plane
    bounds = (0, thin, thin + own, shape[0])
    mu = np.array([1.0, 1.0, 0.5, 0.5], dt)
    lam = mu / dt.type(32.0)
    bes = []
    for r in range(3):
        lay = SlabLayout(shape, r, 3, 2, bounds=bounds)
        be = HipBackend(lay, dt, True, device=0, max_iters=k)
        be.set_params(1.0 / lam, (lam / mu).astype(dt))
        _lib.check(_lib.lib().tvdn_synth_fill(be.code, 4, _lib.shape_arr(shape), synth.SEED_4D, lay.g0 - lay.halo_lo,
                                              lay.local_shape[0], be.orig.data_ptr(), _lib.current_stream(0)))
        be.recon[be.cur].copy_(be.orig)
        bes.append(be)
    big = bes[1]
    assert big.layout.local_shape == (own + 2,) + plane
    assert big.layout.lo_mode == _lib.EDGE_HALO and big.layout.hi_mode == _lib.EDGE_HALO
    grp = LocalSlabs(bes, split_sweeps=True)
    grp.run(k, 0)
    torch.cuda.synchronize()
    lay = big.layout
    orig_rows = big.orig                       # global rows g0-1 .. g1+1
    recon_rows = big.recon_tensor()
    row0 = lay.g0 - 1
    A, B, Cc = plane
    wins = [
        ((lay.g0, 0, 0, 0), (2, 6, 8, 16)),                              # first own row (swept on its own), cube corner
        ((lay.g1 - 2, A - 7, B - 9, Cc - 16), (2, 7, 9, 16)),            # last own row, opposite corner
        ((lay.g0 + own // 2 - 1, A // 2, B // 2 - 3, Cc // 2), (3, 5, 6, 12)),   # interior, across a march seam
        ((lay.g0 + 7, A - 5, 0, Cc - 12), (3, 5, 6, 12)),                # rows 7..9 of the slab: seam between two marches
    ]
    for start, ext in wins:
        # the window's 2k-row halo may reach into the thin slabs: take those rows from them
        halo = 2 * k
        need_lo, need_hi = max(0, start[0] - halo), min(shape[0], start[0] + ext[0] + halo)
        rows_o = torch.cat([be.orig[be.layout.row_lo:be.layout.row_hi] for be in bes], dim=0)[need_lo:need_hi] \
            if (need_lo < row0 or need_hi > row0 + own + 2) else orig_rows[need_lo - row0:need_hi - row0]
        x_row0 = need_lo
        # result rows always come from the big slab
        lo = [need_lo] + [max(0, s - halo) for s in start[1:]]
        hi = [need_hi] + [min(n, s + e + halo) for s, e, n in zip(start[1:], ext[1:], shape[1:])]
        x = rows_o[(slice(0, need_hi - need_lo),) + tuple(slice(a, b) for a, b in zip(lo[1:], hi[1:]))].cpu().numpy().copy()
        ref = oracle.denoise(x, mu, k, True)["recon"]
        inner = tuple(slice(s - a, s - a + e) for s, a, e in zip(start, lo, ext))
        got = recon_rows[(slice(start[0] - row0, start[0] - row0 + ext[0]),)
                         + tuple(slice(s, s + e) for s, e in zip(start[1:], ext[1:]))].cpu().numpy()
        assert bits_equal(got, ref[inner]), (start, ext)
    s = grp.global_sums().cpu().numpy()
    assert np.all(np.isfinite(s)) and np.all(s[:, 2] > 0)
    del grp, bes, big, be, orig_rows, recon_rows, rows_o, got
    gc.collect()
    torch.cuda.empty_cache()


def test_default_path_with_pipelined_transfers_at_full_size(oracle, monkeypatch):
    """denoise4D as a user calls it on the config-2 cube in host memory: tvdn_run shapes its own pipelined transfers (32-row
    chunks, 8 iterations under the upload, 8 over the download, the real helper threads and staging lanes) -- the same bits
    as the plain order, and windows of the result equal the oracle on the enlarged windows of the input."""
    import torch
    import cytvdn_amd as tv
    from cytvdn_amd import _lib, synth
    its = 18
    dt = np.dtype(np.float32)
    mu = np.array([1.0, 1.0, 0.5, 0.5], dt)
    buf = torch.empty(SHAPE, dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().tvdn_synth_fill(0, 4, _lib.shape_arr(SHAPE), synth.SEED_4D, 0, SHAPE[0], buf.data_ptr(),
                                          _lib.current_stream(0)))
    x = buf.cpu().numpy()
    del buf
    torch.cuda.empty_cache()
    monkeypatch.delenv("TVDN_PIPELINE", raising=False)
    assert _lib.pipeline_plan(SHAPE[0], its, x.nbytes) == (32, 8, 8)
    got, bn, dl = tv.denoise4D(x, mu, its, quiet=True)
    monkeypatch.setenv("TVDN_PIPELINE", "0")
    plain, bn0, dl0 = tv.denoise4D(x, mu, its, quiet=True)
    assert hashlib.sha1(got.tobytes()).hexdigest() == hashlib.sha1(plain.tobytes()).hexdigest()
    np.testing.assert_allclose(bn.astype(np.float64), bn0.astype(np.float64), rtol=1e-6)
    np.testing.assert_allclose(dl.astype(np.float64), dl0.astype(np.float64), rtol=1e-6)
    del plain
    halo = 2 * its
    for start, ext in (((0, 0, 0, 0), (6, 6, 8, 8)), ((250, 250, 120, 120), (6, 6, 8, 8)), ((29, 120, 60, 60), (6, 4, 6, 8))):
        lo = [max(0, s - halo) for s in start]
        hi = [min(n, s + e + halo) for s, e, n in zip(start, ext, SHAPE)]
        sl = tuple(slice(a, b) for a, b in zip(lo, hi))
        ref = oracle.denoise(np.ascontiguousarray(x[sl]), mu, its, True)["recon"]
        inner = tuple(slice(s - a, s - a + e) for s, a, e in zip(start, lo, ext))
        assert bits_equal(got[tuple(slice(s, s + e) for s, e in zip(start, ext))], ref[inner]), start


@pytest.mark.parametrize("bc", [2, 0])
def test_streamed_run_at_a_size_that_page_locks_in_place(monkeypatch, bc):
    """The C++ streamed loop on a cube large enough (2 GiB arrays) that `data` and `recon_out` are page-locked IN PLACE
    (csrc/tvdn_stream.hip; the small streamed tests go through staging copies): two passes of four iterations, 16-row
    chunks, against the resident run of the same call -- same bits, same traces; the input is left alone.  Jia-Zhao and
    periodic boundaries (the latter keeps old and new host state apart: 19 pinned arrays)."""
    import torch
    import cytvdn_amd as tv
    from cytvdn_amd import _lib, synth
    shape = (128, 256, 128, 128)
    dt = np.dtype(np.float32)
    mu = np.array([1.0, 1.0, 0.5, 0.5], dt)
    from cytvdn_amd import planner
    pinned = (19 if bc == 0 else 10) * int(np.prod(shape)) * 4          # what the streamed run page-locks
    avail = planner.host_available()
    if avail is not None and pinned + 3 * int(np.prod(shape)) * 4 > 0.8 * avail:
        pytest.skip(f"host offers {avail >> 30} GiB to pin, the run needs {pinned >> 30} GiB")
    buf = torch.empty(shape, dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().tvdn_synth_fill(0, 4, _lib.shape_arr(shape), synth.SEED_4D, 0, shape[0], buf.data_ptr(),
                                          _lib.current_stream(0)))
    x = buf.cpu().numpy()
    del buf
    torch.cuda.empty_cache()
    sha_in = hashlib.sha1(x.tobytes()).hexdigest()
    resident = tv.denoise4D(x, mu, 8, quiet=True, BC_mode=bc)
    monkeypatch.setenv("TVDN_WAVEFRONT", "16,4")
    streamed = tv.denoise4D(x, mu, 8, quiet=True, BC_mode=bc)
    assert hashlib.sha1(x.tobytes()).hexdigest() == sha_in
    assert hashlib.sha1(streamed[0].tobytes()).hexdigest() == hashlib.sha1(resident[0].tobytes()).hexdigest()
    np.testing.assert_allclose(streamed[1].astype(np.float64), resident[1].astype(np.float64), rtol=1e-6)
    np.testing.assert_allclose(streamed[2].astype(np.float64), resident[2].astype(np.float64), rtol=1e-6)


def test_streamed_engine_at_config5_plane_size_against_the_oracle(oracle):
    """The streamed engine at BASELINE config 5's plane size (1024 x 256 x 256 f32, 256 MiB per row; 16 rows = 4 GiB per array: 40 GiB page-locked, inside the 64 GiB the tests may pin)
    DIRECTLY against the oracle: nine FISTA iterations as three passes of three levels, two-row chunks -- (a) every row kept in HBM
    and swept in place, (b) 13 of 16 kept (in place between the streamed rows), (c) every row streamed, the three passes chained,
    recon rebuilt on the device instead of crossing PCIe.  The three results are the same bits, and windows of them at both faces,
    in the middle and across a chunk seam equal the oracle on the enlarged windows of the input."""
    import torch
    from cytvdn_amd import _lib, planner, synth
    from test_gpu_run_streamed import _run
    shape = (16, 1024, 256, 256)
    its, rows, k = 9, 2, 3
    avail = planner.host_available()
    if avail is None or avail < 60 * 2 ** 30:
        pytest.skip("the host offers less than 60 GiB (40 GiB page-locked + the cube and two results)")
    dt = np.dtype(np.float32)
    mu = np.array([1.0, 1.0, 0.5, 0.5], dt)
    buf = torch.empty(shape, dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().tvdn_synth_fill(0, 4, _lib.shape_arr(shape), synth.SEED_4D, 0, shape[0], buf.data_ptr(),
                                          _lib.current_stream(0)))
    x = buf.cpu().numpy()
    del buf
    torch.cuda.empty_cache()
    results = []
    for resident, kind, passes in ((16, 3, 3), (13, 1, 3), (0, 0, 3)):
        st = _lib.RunStats()
        recon, sums, _, ran = _run(x, mu, its, 0, stream=(rows, k), resident=resident, stats=st)
        assert ran == its and st.engine == 1 and (st.stream_rows, st.stream_k, st.n_passes) == (rows, k, passes)
        assert st.resident_rows == resident and st.kept_in_place == kind
        if resident == 0:       # recon stays on the device between the passes: data term up 3 x, state 2 x each way, recon down once
            assert st.h2d_bytes == x.nbytes * (3 + 2 * 8) and st.d2h_bytes == x.nbytes * (1 + 3 * 8)
        results.append((hashlib.sha1(recon.tobytes()).hexdigest(), sums.copy()))
        if resident == 16:
            kept = recon
        del recon
    assert results[0][0] == results[1][0] == results[2][0]
    np.testing.assert_allclose(results[1][1], results[0][1], rtol=1e-12)
    np.testing.assert_allclose(results[2][1], results[0][1], rtol=1e-12)
    recon = kept
    halo = 2 * its
    A, B, Cc = shape[1:]
    for start, ext in (((0, 0, 0, 0), (3, 6, 8, 16)), ((13, A - 6, B - 9, Cc - 16), (3, 6, 9, 16)), ((7, 500, 100, 64), (3, 5, 6, 12)),
                       ((5, A - 7, 0, Cc - 12), (2, 7, 8, 12))):
        lo = [max(0, s - halo) for s in start]
        hi = [min(n, s + e + halo) for s, e, n in zip(start, ext, shape)]
        ref = oracle.denoise(np.ascontiguousarray(x[tuple(slice(a, b) for a, b in zip(lo, hi))]), mu, its, True)["recon"]
        inner = tuple(slice(s - a, s - a + e) for s, a, e in zip(start, lo, ext))
        assert bits_equal(recon[tuple(slice(s, s + e) for s, e in zip(start, ext))], ref[inner]), start
    # the sums of a cube this size are finite and positive
    assert np.all(np.isfinite(results[0][1])) and np.all(results[0][1][:, 2] > 0)
