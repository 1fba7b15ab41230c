"""tvdn_run's pipelined transfers (csrc/tvdn_run.hip): a resident denoise3D/4D from NumPy whose first iterations run under
the upload and whose last ones run over the download (a wavefront of partial sweeps on the resident arrays) gives the bits
of the plain loop, i.e. the oracle's (reference loop: cyTVDN/cyTVDN.py:148-242) -- recon and the b_norm / delta_recon
traces -- for every chunk height and depth, FISTA / unaccelerated / hybrid schedules (a transition inside the start or the
end wavefront), f32 / f64, 3-D / 4-D, scalar packs; through the Python API and through the C entry point directly."""
import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,dtype,its,fista,pipe", [
    ((40, 4, 8, 16), np.float32, 12, True, "8,4,4"),
    ((40, 4, 8, 16), np.float32, 12, True, "5,6,6"),          # chunks that do not divide the cube, no middle part
    ((23, 6, 16), np.float64, [5, 4], True, "4,3,3"),         # hybrid: the d -> b transition in the middle
    ((23, 6, 16), np.float64, [2, 7], True, "4,4,3"),         # ... inside the start wavefront
    ((23, 6, 16), np.float32, [8, 2], True, "4,3,4"),         # ... inside the end wavefront
    ((19, 3, 5, 7), np.float32, 9, False, "3,2,5"),           # unaccelerated, scalar packs
    ((17, 3, 4, 8), np.float32, 6, True, "32,3,3"),           # one chunk taller than the cube
    ((33, 3, 4, 8), np.float32, 7, True, "1,3,2"),            # one-row chunks
    ((16, 3, 4, 8), np.float32, 5, True, "4,5,0"),            # no end wavefront: whole-cube download
    ((16, 3, 4, 8), np.float32, 4, True, "4,1,3"),
])
def test_pipelined_run_matches_the_oracle(oracle, monkeypatch, shape, dtype, its, fista, pipe):
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=29, dtype=dt) + dt.type(0.25)
    x0 = x.copy()
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    ref = oracle.denoise(x, mu, its, fista)
    monkeypatch.setenv("TVDN_PIPELINE", pipe)
    recon, bn, dl = fn(x, mu, its, FISTA=fista, quiet=True)
    assert bits_equal(x, x0)
    assert bits_equal(recon, ref["recon"])
    np.testing.assert_allclose(bn.astype(np.float64), ref["b_norm64"].astype(dt).astype(np.float64),
                               rtol=1e-6 if dt == np.float32 else 1e-9)
    want = (ref["delta64"].astype(dt) / ref["rnorm64"].astype(dt)).astype(np.float64)
    np.testing.assert_allclose(dl.astype(np.float64), want, rtol=1e-6 if dt == np.float32 else 1e-9)
    monkeypatch.setenv("TVDN_PIPELINE", "0")
    plain = fn(x, mu, its, FISTA=fista, quiet=True)
    assert bits_equal(recon, plain[0])
    # the same f64 sums, folded per partial launch instead of per sweep: the last bits may differ
    np.testing.assert_allclose(bn.astype(np.float64), plain[1].astype(np.float64), rtol=1e-6 if dt == np.float32 else 1e-12)


@pytest.mark.parametrize("shape,dtype,n_f,n_p,pipe", [
    ((40, 4, 8, 16), np.float32, 12, 0, "8,4,4"),
    ((23, 6, 16), np.float64, 2, 7, "4,4,3"),
    ((19, 3, 5, 7), np.float32, 0, 9, "3,2,5"),
    ((33, 3, 4, 8), np.float32, 7, 0, "1,3,2"),
    ((16, 3, 4, 8), np.float32, 5, 0, "4,0,5"),            # no start wavefront: whole-cube upload
    ((16, 3, 4, 8), np.float32, 6, 0, "4,9,9"),            # more levels asked for than iterations
])
def test_tvdn_run_pipelined_against_the_oracle_directly(oracle, monkeypatch, shape, dtype, n_f, n_p, pipe):
    """The C entry point itself (no Python driver in between), sums against the oracle's f64 yardsticks."""
    from test_gpu_run_streamed import _check_traces, _oracle, _run
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    x = synth.cube(shape, seed=31, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:len(shape)], dt)
    ref = _oracle(oracle, x, mu, n_f, n_p)
    monkeypatch.setenv("TVDN_PIPELINE", pipe)
    recon, sums, _, ran = _run(x, mu, n_f, n_p)
    assert ran == n_f + n_p and bits_equal(recon, ref["recon"])
    _check_traces(sums, ref, n_f + n_p)
    recon2 = x.copy()                                          # in place: data and recon_out the same array
    got = _run_in_place(recon2, mu, n_f, n_p)
    assert bits_equal(got, ref["recon"])


def _run_in_place(x, mu, n_f, n_p):
    import ctypes as C
    from cytvdn_amd import _lib
    dt, nd = x.dtype, x.ndim
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    a = _lib.RunArgs(dtype=_lib.dtype_code(dt), ndim=nd, bc_mode=2, device=0, n_fista=n_f, n_plain=n_p)
    for i, s in enumerate(x.shape):
        a.shape[i] = s
    for q in range(nd):
        a.clip[q] = float((1.0 / lam)[q])
        a.lambda_mu[q] = float((lam / mu).astype(dt)[q])
    sums = np.zeros((n_f + n_p, 3))
    a.data = a.recon_out = x.ctypes.data
    a.sums_out = sums.ctypes.data
    _lib.check(_lib.lib().tvdn_run(C.byref(a)))
    return x


def test_progress_callback_counts_every_slot(monkeypatch):
    """tvdn_run_args.progress: monotone slot counts ending at n_fista + n_plain, plain order and pipelined; the
    unaccelerated phase counts on from n_fista after an early FISTA stop."""
    import ctypes as C
    from cytvdn_amd import _lib, synth
    dt = np.dtype(np.float32)
    x = synth.cube((24, 3, 4, 8), seed=3, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6], dt)
    lam = mu / dt.type(32.0)

    def run(n_f, n_p, stop=None):
        seen = []
        hook = C.CFUNCTYPE(None, C.c_int32, C.c_void_p)(lambda n, _u: seen.append(int(n)))
        a = _lib.RunArgs(dtype=0, ndim=4, bc_mode=2, device=0, n_fista=n_f, n_plain=n_p, use_stop=int(stop is not None),
                         stop=float(stop or 0.0))
        for i, s in enumerate(x.shape):
            a.shape[i] = s
        for q in range(4):
            a.clip[q] = float((1.0 / lam)[q])
            a.lambda_mu[q] = float((lam / mu).astype(dt)[q])
        recon = np.empty_like(x)
        sums = np.zeros((n_f + n_p, 3))
        a.data, a.recon_out, a.sums_out = x.ctypes.data, recon.ctypes.data, sums.ctypes.data
        a.progress = C.cast(hook, C.c_void_p)
        _lib.check(_lib.lib().tvdn_run(C.byref(a)))
        return seen

    monkeypatch.setenv("TVDN_PIPELINE", "0")
    assert run(5, 3) == list(range(1, 9))
    monkeypatch.setenv("TVDN_PIPELINE", "4,3,2")
    assert run(5, 3) == [3, 4, 5, 6, 8]
    monkeypatch.setenv("TVDN_PIPELINE", "0")
    seen = run(6, 4, stop=1e30)                                # both phases stop after their first iteration
    assert seen == [1, 7]


def test_a_nonfinite_first_row_takes_the_plain_order(oracle, monkeypatch):
    """The wavefront's top face uses the Jia-Zhao zero, which upstream computes only while row 0 is finite
    (anisotropic.pyx:65-73): such a cube is not pipelined, and still equals the oracle."""
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(np.float32)
    x = synth.cube((24, 3, 4, 8), seed=5, dtype=dt) + dt.type(0.25)
    x[0, 1, 2, 3] = np.inf
    mu = np.array([1.0, 0.8, 0.5, 0.6], dt)
    monkeypatch.setenv("TVDN_PIPELINE", "4,3,3")
    got = tv.denoise4D(x, mu, 8, quiet=True)
    ref = oracle.denoise(x, mu, 8, True)
    assert np.isnan(ref["recon"][-1]).any() and bits_equal(got[0], ref["recon"])


def test_upstream_progress_bars_are_fed_from_the_library(oracle, capfd, monkeypatch):
    """quiet=False: the two tqdm bars of cyTVDN.py:148-151 / :196-199 advance from tvdn_run's callback and end at their
    totals; the results are those of the quiet call."""
    pytest.importorskip("tqdm")
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(np.float32)
    x = synth.cube((24, 3, 4, 8), seed=11, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6], dt)
    monkeypatch.setenv("TVDN_PIPELINE", "4,3,2")
    loud = tv.denoise4D(x, mu, [6, 4], FISTA=True, quiet=False)
    err = capfd.readouterr().err
    assert "FISTA Accelerated TV Denoising: 100%" in err and "6/6" in err
    assert "Unaccelerated TV Denoising: 100%" in err and "4/4" in err
    quiet = tv.denoise4D(x, mu, [6, 4], FISTA=True, quiet=True)
    ref = oracle.denoise(x, mu, [6, 4], True)
    assert bits_equal(loud[0], ref["recon"]) and bits_equal(quiet[0], ref["recon"])
    for u, v in zip(loud[1:], quiet[1:]):
        assert bits_equal(u, v)


@pytest.mark.parametrize("shape,dtype,n_f,n_p,pipe", [
    ((24, 3, 4, 8), np.float32, 7, 0, "0"), ((24, 3, 4, 8), np.float32, 5, 3, "4,3,2"), ((13, 6, 16), np.float64, 0, 6, "0"),
])
def test_tvdn_run_in_the_callers_workspace(oracle, monkeypatch, shape, dtype, n_f, n_p, pipe):
    """tvdn_run_args.workspace: the state in device memory the caller brings (here a torch block full of 0xFF bytes, i.e.
    NaNs: whatever the run needs zeroed it zeroes itself) -- the oracle's bits; too small or misaligned is refused."""
    import ctypes as C
    import torch
    from test_gpu_run_streamed import _check_traces, _oracle
    from cytvdn_amd import _lib, synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=37, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    ref = _oracle(oracle, x, mu, n_f, n_p)
    monkeypatch.setenv("TVDN_PIPELINE", pipe)
    a = _lib.RunArgs(dtype=_lib.dtype_code(dt), ndim=nd, bc_mode=2, device=0, n_fista=n_f, n_plain=n_p)
    for i, s in enumerate(shape):
        a.shape[i] = s
    for q in range(nd):
        a.clip[q] = float((1.0 / lam)[q])
        a.lambda_mu[q] = float((lam / mu).astype(dt)[q])
    need = C.c_int64()
    _lib.check(_lib.lib().tvdn_run_workspace_bytes(C.byref(a), C.byref(need)))
    n_arr = 3 + nd * (3 if n_f else 2)
    assert need.value == n_arr * (-(-x.nbytes // 256) * 256 + 4096)
    ws = torch.full((need.value + 256,), 0xFF, dtype=torch.uint8, device="cuda")
    recon, sums = np.empty_like(x), np.zeros((n_f + n_p, 3))
    a.data, a.recon_out, a.sums_out = x.ctypes.data, recon.ctypes.data, sums.ctypes.data
    a.workspace, a.workspace_bytes = ws.data_ptr(), need.value
    for _ in range(2):                                         # the second run finds the first one's leftovers
        _lib.check(_lib.lib().tvdn_run(C.byref(a)))
        assert bits_equal(recon, ref["recon"])
        _check_traces(sums, ref, n_f + n_p)
    a.workspace_bytes = need.value - 1
    assert _lib.lib().tvdn_run(C.byref(a)) == -1 and b"workspace" in _lib.lib().tvdn_last_error()
    a.workspace, a.workspace_bytes = ws.data_ptr() + 64, need.value
    assert _lib.lib().tvdn_run(C.byref(a)) == -1 and b"aligned" in _lib.lib().tvdn_last_error()
    host = np.zeros(need.value + 512, np.uint8)                # host memory is refused, not dereferenced by a kernel
    a.workspace = (host.ctypes.data + 255) // 256 * 256
    assert _lib.lib().tvdn_run(C.byref(a)) == -1 and b"not device memory" in _lib.lib().tvdn_last_error()


def test_the_binding_shown_in_integration_md_runs(oracle):
    """the ctypes stub a cyTVDN maintainer would add around tvdn_run, executed as printed (only the
    library's path filled in) -- the oracle's reconstruction, a progress bar that ends at its total."""
    import os
    import re
    from cytvdn_amd import _lib, synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# cyTVDN/_hip\.py  \(new file upstream\)\nimport ctypes as C\nL = C\.CDLL.*?)```", text, re.S)
    assert m
    ns = {"np": np}
    exec(m.group(1).replace('"libtvdn_hip.so"', repr(_lib.LIB_PATH)), ns)
    dt = np.dtype(np.float32)
    x = synth.cube((20, 3, 4, 8), seed=41, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6], dt)
    lam = mu / dt.type(32.0)

    class Bar:
        n = 0

        def update(self, k):
            self.n += k

    bar = Bar()
    recon, b_norm, delta = ns["denoise4D_hip"](x, 1.0 / lam, (lam / mu).astype(dt), 6, 3, 2, bar)
    ref = oracle.denoise(x, mu, [6, 3], True)
    assert bits_equal(recon, ref["recon"]) and bar.n == 9
    np.testing.assert_allclose(b_norm.astype(np.float64), ref["b_norm64"].astype(dt).astype(np.float64), rtol=1e-6)
    np.testing.assert_allclose(delta.astype(np.float64), (ref["delta64"].astype(dt) / ref["rnorm64"].astype(dt)).astype(np.float64), rtol=1e-6)


@pytest.mark.parametrize("shape,its", [((24, 6, 16, 32), 9), ((64, 32, 128, 256), 8)])
def test_concurrent_calls_from_threads(shape, its):
    """Three Python threads denoise three different cubes at once (the GIL is released inside tvdn_run; every call has its
    own context, streams and workspace; the staging lanes are shared behind a mutex): each result equals the serial one.
    The second shape is large enough (256 MiB) for the pipelined transfers with their helper threads."""
    import hashlib
    import threading
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    mu = np.array([1.0, 1.0, 0.5, 0.5], np.float32)

    def sha(a):
        return hashlib.sha1(a.tobytes()).hexdigest()

    xs = [synth.cube(shape, seed=s, dtype=np.float32) + np.float32(0.25) for s in (1, 2, 3)]
    want = [sha(tv.denoise4D(x, mu, its, quiet=True)[0]) for x in xs]
    got = [[None] * 3 for _ in xs]
    errors = []

    def work(i):
        try:
            for r in range(3):
                got[i][r] = sha(tv.denoise4D(xs[i], mu, its, quiet=True)[0])
        except Exception as e:      # surfaces in the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(xs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert all(g == w for gs, w in zip(got, want) for g in gs)


def test_concurrent_calls_that_audition(monkeypatch):
    """The same with runs that audition their placement (TVDN_AUDITION=3; by default from 400 iterations on): the probes of
    several threads time sweeps with HIP events on a reduction context -- each audition has its own and they take turns per
    device (driver._audition_lock), so nothing races on the process-wide context's event list or scratch."""
    import hashlib
    import threading
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    monkeypatch.setenv("TVDN_AUDITION", "3")
    mu = np.array([1.0, 1.0, 0.5, 0.5], np.float32)
    xs = [synth.cube((24, 6, 16, 32), seed=s, dtype=np.float32) + np.float32(0.25) for s in (4, 5, 6)]
    want = [hashlib.sha1(tv.denoise4D(x, mu, 7, quiet=True)[0].tobytes()).hexdigest() for x in xs]
    got = [[None] * 3 for _ in xs]
    errors = []

    def work(i):
        try:
            for r in range(3):
                got[i][r] = hashlib.sha1(tv.denoise4D(xs[i], mu, 7, quiet=True)[0].tobytes()).hexdigest()
        except Exception as e:
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(xs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert all(g == w for gs, w in zip(got, want) for g in gs)


def test_run_stats_say_where_the_time_and_bytes_went(oracle, monkeypatch):
    """tvdn_run_args.stats (ABI 6): engine, set-up / loop / total seconds, bytes across PCIe, the audition's probe times."""
    import ctypes as C
    from cytvdn_amd import _lib, synth
    monkeypatch.setenv("TVDN_AUDITION", "3")
    monkeypatch.setenv("TVDN_KEEP_STATE", "0")
    x = synth.cube((24, 6, 16, 32), seed=11, dtype=np.float32) + np.float32(0.25)
    mu = np.array([1.0, 1.0, 0.5, 0.5], np.float32)
    lam = mu / np.float32(32.0)
    recon, sums = np.empty_like(x), np.zeros((5, 3))
    st = _lib.RunStats()
    a = _lib.RunArgs(dtype=0, ndim=4, bc_mode=2, device=0, n_fista=5, n_plain=0)
    for i, v in enumerate(x.shape):
        a.shape[i] = v
    for q in range(4):
        a.clip[q], a.lambda_mu[q] = float((1.0 / lam)[q]), float((lam / mu).astype(np.float32)[q])
    a.data, a.recon_out, a.sums_out, a.stats = x.ctypes.data, recon.ctypes.data, sums.ctypes.data, C.addressof(st)
    _lib.check(_lib.lib().tvdn_run(C.byref(a)))
    assert bits_equal(recon, oracle.denoise(x, mu, 5, True)["recon"])
    d = st.as_dict()
    assert d["engine"] == 0 and d["h2d_bytes"] == x.nbytes and d["d2h_bytes"] == x.nbytes
    assert d["audition_n"] == 3 and len(d["audition_ms"]) == 3 and 0 <= d["audition_kept"] < 3 and min(d["audition_ms"]) > 0
    assert 0 < d["setup_s"] and 0 < d["loop_s"] and d["setup_s"] + d["loop_s"] <= d["total_s"] * 1.001
    # streamed: the passes' bytes; 5 iterations at k = 2 are three passes of 10 arrays up (the first with both d arrays) and 9 down
    a.stream_rows, a.stream_k = 4, 2
    _lib.check(_lib.lib().tvdn_run(C.byref(a)))
    assert bits_equal(recon, oracle.denoise(x, mu, 5, True)["recon"])
    d = st.as_dict()
    assert d["engine"] == 1 and d["stream_rows"] == 4 and d["stream_k"] == 2 and d["n_passes"] == 3
    assert 0 < d["h2d_bytes"] <= 3 * 10 * x.nbytes and 0 < d["d2h_bytes"] <= 3 * 9 * x.nbytes
    assert d["loop_s"] > 0 and d["total_s"] >= d["loop_s"]


def test_the_state_block_is_kept_between_runs_without_a_workspace(oracle, monkeypatch):
    """A C caller that brings no workspace: tvdn_run keeps the state block of its last resident run (one per device) for the
    next one -- visible as HBM that stays in use after the call and comes back with tvdn_release_cache() --, reuses it for
    a run that fits, replaces it for one that does not, never keeps one under TVDN_KEEP_STATE=0; the results are the oracle's."""
    import torch
    from test_gpu_run_streamed import _oracle, _run
    from cytvdn_amd import _lib, synth
    L = _lib.lib()
    dt = np.dtype(np.float32)
    mu = np.array([1.0, 0.8, 0.5, 0.6], dt)
    big = synth.cube((64, 16, 64, 64), seed=5, dtype=dt) + dt.type(0.25)        # 16 MiB arrays: 240 MiB of state
    small = synth.cube((8, 4, 8, 16), seed=6, dtype=dt) + dt.type(0.25)
    assert L.tvdn_release_cache() == 0
    torch.cuda.empty_cache()

    def free():
        torch.cuda.synchronize()
        return int(torch.cuda.mem_get_info(0)[0])

    f0 = free()
    ref_big, ref_small = _oracle(oracle, big, mu, 3, 0), _oracle(oracle, small, mu, 3, 2)
    assert bits_equal(_run(big, mu, 3, 0)[0], ref_big["recon"])
    held = f0 - free()
    assert 200 << 20 <= held <= 320 << 20                                        # the 15 x 16 MiB state is still allocated
    assert bits_equal(_run(big, mu, 3, 0)[0], ref_big["recon"])                  # ... and serves the next run
    assert abs((f0 - free()) - held) <= 8 << 20
    assert bits_equal(_run(small, mu, 3, 2)[0], ref_small["recon"])              # too large for this one: replaced
    assert f0 - free() <= 64 << 20
    assert L.tvdn_release_cache() == 0 and f0 - free() <= 8 << 20
    monkeypatch.setenv("TVDN_KEEP_STATE", "0")
    assert bits_equal(_run(big, mu, 3, 0)[0], ref_big["recon"])
    assert f0 - free() <= 8 << 20


@pytest.mark.parametrize("lanes", [False, True], ids=["page-locked-result", "lanes"])
def test_result_rows_go_straight_into_a_page_locked_result_array(oracle, monkeypatch, lanes):
    """The pipelined download: a helper page-locks the caller's result array where it is while the middle iterations run, and the
    finished rows cross PCIe straight into it (arrays of 256 MiB and more; the threshold lowered here); TVDN_RESULT_LANES=1 keeps
    the pinned lanes.  Same bits either way, the input untouched, also when the result array IS the input."""
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    monkeypatch.setenv("TVDN_PIN_IN_PLACE_MIN", "64K")
    monkeypatch.setenv("TVDN_PIPELINE", "8,3,4")
    if lanes:
        monkeypatch.setenv("TVDN_RESULT_LANES", "1")
    dt = np.dtype(np.float32)
    shape = (40, 8, 32, 64)
    x = synth.cube(shape, seed=77, dtype=dt) + dt.type(0.25)
    keep = x.copy()
    mu = np.array([1.0, 0.8, 0.5, 0.6], dt)
    got = tv.denoise4D(x, mu, [7, 4], quiet=True)
    ref = oracle.denoise(keep, mu, [7, 4], True)
    assert bits_equal(x, keep) and bits_equal(got[0], ref["recon"])
    np.testing.assert_allclose(got[1].astype(np.float64), ref["b_norm64"].astype(dt).astype(np.float64), rtol=1e-6)
