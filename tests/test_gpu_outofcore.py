"""Out-of-core (host-resident, temporally blocked) runs through denoise3D/4D against the in-core run and the oracle:
bit-identical recon, identical scalar traces, for every mix of chunk height and iterations-per-pass."""
import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,dtype,its,fista,rows,k,with_ref", [
    ((23, 3, 4, 8), "float32", 9, True, 5, 3, False),
    ((23, 3, 4, 8), "float32", 9, True, 5, 1, False),
    ((23, 3, 4, 8), "float32", [5, 4], True, 4, 4, True),     # hybrid: the d -> b transition inside a pass
    ((17, 6, 16), "float64", 7, False, 3, 5, False),
    ((17, 6, 16), "float64", 7, True, 17, 2, True),            # a single block = the whole cube
    ((9, 2, 5, 7), "float32", 6, True, 2, 8, False),           # halo deeper than the cube: every block is clipped
    ((40, 3, 4, 8), "float32", 11, True, 7, 4, False),
])
def test_staged_equals_in_core(oracle, monkeypatch, shape, dtype, its, fista, rows, k, with_ref):
    """TVDN_STAGED forces a streamed run of these (rows, k) on a cube that fits: the library's loop (tvdn_run,
    csrc/tvdn_stream.hip)."""
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=57, dtype=dt) + dt.type(0.25)
    refd = synth.cube(shape, seed=57, dtype=dt, kind="mean") if with_ref else None
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    want = fn(x, mu, its, FISTA=fista, reference_data=refd, quiet=True)
    monkeypatch.setenv("TVDN_STAGED", f"{rows},{k}")
    got = fn(x, mu, its, FISTA=fista, reference_data=refd, quiet=True)
    assert len(got) == len(want)
    assert bits_equal(got[0], want[0])
    ref = oracle.denoise(x, mu, its, fista, reference_data=refd)
    assert bits_equal(got[0], ref["recon"])
    for a, b in zip(got[1:], want[1:]):
        np.testing.assert_allclose(a, b, rtol=1e-6 if dt == np.float32 else 1e-12)


def test_staged_early_stop_matches(monkeypatch):
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    x = synth.cube((12, 5, 8, 12), seed=14, dtype=np.float32)
    mu = np.array([1.0, 1.0, 0.5, 0.5], np.float32)
    want = tv.denoise4D(x, mu, [30, 6], stopping_relative_change=0.03, quiet=True)
    monkeypatch.setenv("TVDN_STAGED", "5,8")
    got = tv.denoise4D(x, mu, [30, 6], stopping_relative_change=0.03, quiet=True)
    assert bits_equal(got[0], want[0])
    assert np.array_equal(got[2] == 0, want[2] == 0)
    np.testing.assert_allclose(got[2], want[2], rtol=1e-6)


@pytest.mark.parametrize("shape,dtype,its,fista,rows,k,bc", [
    ((23, 3, 4, 8), "float32", 9, True, 5, 3, 0),             # periodic BC: the cube's two ends are each other's halo
    ((17, 6, 16), "float64", [4, 3], True, 4, 6, 0),
    ((6, 3, 4, 8), "float32", 7, True, 2, 9, 0),              # k capped at the cube height
    ((15, 3, 4, 8), "float32", 8, True, 1, 5, 0),             # one-row chunks
    ((12, 5, 8, 12), "float32", 6, False, 3, 2, 0),           # unaccelerated
])
def test_wavefront_periodic(oracle, monkeypatch, shape, dtype, its, fista, rows, k, bc):
    """Periodic boundaries beyond HBM, by the library's streamed loop (csrc/tvdn_stream.hip: virtual cube between k
    wrapped rows, old and new host state apart): the oracle's bits."""
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=57, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    monkeypatch.setenv("TVDN_WAVEFRONT", f"{rows},{k}")
    got = fn(x, mu, its, FISTA=fista, BC_mode=bc, quiet=True)
    ref = oracle.denoise(x, mu, its, fista, BC_mode=bc)
    assert bits_equal(got[0], ref["recon"])
    np.testing.assert_allclose(got[1].astype(np.float64), ref["b_norm64"].astype(dt).astype(np.float64),
                               rtol=1e-6 if dt == np.float32 else 1e-12)


@pytest.mark.parametrize("shape,dtype,its,fista,rows,k", [
    ((23, 3, 4, 8), "float32", 9, True, 5, 3),
    ((23, 3, 4, 8), "float32", 9, True, 2, 9),               # deeper than a chunk: many levels per row
    ((23, 3, 4, 8), "float32", [5, 4], True, 4, 4),           # hybrid: the d -> b transition inside a pass
    ((23, 3, 4, 8), "float32", [5, 4], True, 3, 9),           # ... and inside a single pass
    ((17, 6, 16), "float64", 7, False, 3, 5),
    ((17, 6, 16), "float64", 7, True, 17, 2),
    ((9, 2, 5, 7), "float32", 6, True, 2, 8),                 # more levels than rows in the cube
    ((40, 3, 4, 8), "float32", 11, True, 7, 4),
    ((5, 3, 4, 8), "float32", 12, True, 16, 5),               # chunk taller than the cube
    ((19, 3, 4, 8), "float32", [6, 3], True, 1, 7),           # one-row chunks: rings of three rows per level
    ((13, 5, 12), "float64", 8, True, 1, 4),
])
def test_wavefront_equals_in_core(oracle, monkeypatch, shape, dtype, its, fista, rows, k):
    """The wavefront (parallelogram) schedule: every row of every iteration level computed once, still bit-identical
    (recon and the b_norm / delta_recon / MSE traces) -- the library's loop (tvdn_run, stream_rows / stream_k)."""
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=57, dtype=dt) + dt.type(0.25)
    refd = synth.cube(shape, seed=57, dtype=dt, kind="mean") if (rows + k) % 2 else None
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    want = fn(x, mu, its, FISTA=fista, reference_data=refd, quiet=True)
    monkeypatch.setenv("TVDN_WAVEFRONT", f"{rows},{k}")
    got = fn(x, mu, its, FISTA=fista, reference_data=refd, quiet=True)
    assert len(got) == len(want)
    assert bits_equal(got[0], want[0])
    ref = oracle.denoise(x, mu, its, fista)
    assert bits_equal(got[0], ref["recon"])
    for a, b in zip(got[1:], want[1:]):
        np.testing.assert_allclose(a, b, rtol=1e-6 if dt == np.float32 else 1e-12)


@pytest.mark.parametrize("stop", [None, 0.05], ids=["no-stop-rule", "stop-rule"])
def test_planner_picks_the_streamed_engines_by_itself(oracle, monkeypatch, stop):
    """with the HBM the planner may count on capped (TVDN_HBM_LIMIT) below the 39 MB this cube's state
    needs, denoise4D must choose the out-of-core engine on its own and still return the oracle's bits: the library's
    streamed loop (tvdn_run with stream_rows / stream_k from tvdn_stream_plan; one iteration per pass with a stopping rule)."""
    import cytvdn_amd as tv
    from cytvdn_amd import driver, synth
    calls = []
    real_dl = driver._run_device_list

    def spy(devices, *a, **k):
        if k.get("stream") is not None:
            calls.append(("library", k["stream"]))
        return real_dl(devices, *a, **k)

    monkeypatch.setattr(driver, "_run_device_list", spy)
    monkeypatch.delenv("TVDN_WAVEFRONT", raising=False)
    monkeypatch.delenv("TVDN_STAGED", raising=False)
    monkeypatch.setenv("TVDN_HBM_LIMIT", "24M")
    shape, dt = (40, 8, 32, 64), np.dtype(np.float32)
    x = synth.cube(shape, seed=5, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 1.0, 0.5, 0.5], dt)
    its = [5, 3] if stop is None else 30
    got = tv.denoise4D(x, mu, its, FISTA=True, stopping_relative_change=stop, quiet=True)
    ref = oracle.denoise(x, mu, its, True, stopping_relative_change=stop)
    assert calls and calls[0][0] == "library", calls
    if stop is not None:
        assert 0 < np.count_nonzero(got[2]) < 30                     # the rule did fire
    assert bits_equal(got[0], ref["recon"])
    assert np.array_equal(got[2] == 0, ref["delta_recon"] == 0)
    # the same call with room to spare stays in core and gives the same bits
    calls.clear()
    monkeypatch.setenv("TVDN_HBM_LIMIT", "4G")
    again = tv.denoise4D(x, mu, its, FISTA=True, stopping_relative_change=stop, quiet=True)
    assert not calls and bits_equal(again[0], got[0])


def test_a_cube_between_resident_and_streamed_keeps_every_row_in_place(oracle, monkeypatch):
    """A cube whose 15-array state does not fit the HBM the planner may count on, but whose 10 kept arrays and a few rings do (up
    to 1.36 x what fits resident): the library's plan for that much HBM (tvdn_stream_plan, what denoise4D asks for) keeps every
    row in HBM on the lean layout and sweeps it in place -- nothing page-locked, nothing across PCIe between the passes -- and
    denoise4D gets there by itself.  The oracle's bits either way; plan_run says the same beforehand."""
    import ctypes as C
    import cytvdn_amd as tv
    from cytvdn_amd import _lib, driver, planner, synth
    monkeypatch.delenv("TVDN_WAVEFRONT", raising=False)
    monkeypatch.delenv("TVDN_STAGED", raising=False)
    monkeypatch.setenv("TVDN_HBM_LIMIT", "36M")            # 39 MB of resident state; 26 MB of kept arrays
    shape, dt = (40, 8, 32, 64), np.dtype(np.float32)
    x = synth.cube(shape, seed=6, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 1.0, 0.5, 0.5], dt)
    lam = mu / dt.type(32.0)
    its = 12
    plan = planner.plan_run(shape, dt, True, 1)
    assert plan["mode"] == "wavefront" and plan["resident_rows_per_rank"] == 40 and plan["host_bytes_per_rank"] == 0, plan
    assert plan["bytes_per_gpu"] <= 0.85 * 36 * 2 ** 20
    ref = oracle.denoise(x, mu, its, True)
    rows, k = driver._library_stream_plan(x, its, 0, False, False, 2, 0, plan["hbm_bytes"])
    assert rows >= 2 and 3 <= k <= its
    a = _lib.RunArgs(dtype=0, ndim=4, bc_mode=2, device=0, n_fista=its, n_plain=0, stream_rows=rows, stream_k=k, stream_resident=-1)
    for i, v in enumerate(shape):
        a.shape[i] = v
    for q in range(4):
        a.clip[q] = float((1.0 / lam)[q])
        a.lambda_mu[q] = float((lam / mu).astype(dt)[q])
    recon, sums, st = np.empty_like(x), np.zeros((its, 3)), _lib.RunStats()
    a.data, a.recon_out, a.sums_out, a.stats = x.ctypes.data, recon.ctypes.data, sums.ctypes.data, C.addressof(st)
    _lib.check(_lib.lib().tvdn_run(C.byref(a)))
    assert st.engine == 1 and st.resident_rows == 40 and st.kept_in_place == 3 and (st.stream_rows, st.stream_k) == (rows, k), (st.engine, st.resident_rows, st.kept_in_place, st.stream_k)
    assert st.h2d_bytes == x.nbytes and st.d2h_bytes == x.nbytes        # the cube in, the result out
    assert bits_equal(recon, ref["recon"])
    np.testing.assert_allclose(sums[:, 0], ref["b_norm64"], rtol=1e-9)
    got = tv.denoise4D(x, mu, its, quiet=True)
    assert bits_equal(got[0], ref["recon"])


def test_planner_counts_what_torchs_cache_holds():
    """A process that has just denoised a cube keeps its state block in torch's caching allocator: the planner must count
    that as available (the next resident run reuses it), else the second large cube of a process is streamed for nothing."""
    import torch
    from cytvdn_amd import planner
    torch.cuda.empty_cache()
    base = planner.hbm_available(0)
    t = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
    held = planner.hbm_available(0)
    del t                                                       # back to torch's cache, not to the driver
    free_now = int(torch.cuda.mem_get_info(0)[0])
    cached = planner.hbm_available(0)
    assert held <= base - (2 << 30) + (64 << 20)
    assert cached >= free_now + (2 << 30) - (64 << 20) and abs(cached - base) <= (64 << 20)
    torch.cuda.empty_cache()


def test_device_list_streams_its_slabs_when_told_to(oracle, monkeypatch):
    """denoise4D(device=[0, 0, 0]) with TVDN_WAVEFRONT=rows,k: three slabs, each STREAMED through its device from host arrays
    they share (tvdn_run with a device list and stream_rows / stream_k; BASELINE configs[4] in structure) -- the oracle's bits
    and traces, hybrid schedule included."""
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(np.float32)
    x = synth.cube((26, 4, 8, 16), seed=61, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6], dt)
    monkeypatch.setenv("TVDN_WAVEFRONT", "3,4")
    got = tv.denoise4D(x, mu, [6, 3], quiet=True, device=[0, 0, 0])
    ref = oracle.denoise(x, mu, [6, 3], True)
    assert bits_equal(got[0], ref["recon"])
    np.testing.assert_allclose(got[1].astype(np.float64), ref["b_norm64"].astype(dt).astype(np.float64), rtol=1e-6)
