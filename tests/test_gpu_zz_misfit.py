"""tvdn_plan's arithmetic and tvdn_run's refusals of cubes that cannot fit -- LAST file of the `-m gpu` suite on purpose
(the name sorts after every other test file), because this is the one test that hands the library a shape far beyond
the buffers behind it and relies on being refused.  In round 2 a version of it met a library that did not refuse and
took two GPU boxes down; hence the belt and braces here:
  * TVDN_HOST_LIMIT=1G for the whole module: whatever the library decides, it may not page-lock more than 0.8 GiB;
  * `data` and `recon_out` are real, distinct, MiB-sized arrays (a library that wrongly read a row would fault inside
    this process's own heap mapping rather than walk off a 16-byte array);
  * the refusal of the host side is checked FIRST through tvdn_stream_host_need (pure arithmetic, also tested without
    a GPU in tests/test_host_guard_cpu.py); tvdn_run is only called when that says it will be refused.
Replaces nothing upstream: cyTVDN's check_memory (cyTVDN/cyTVDN.py:438-467) only prints."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BIG = (4096, 512, 256, 256)     # 8 TiB of state: no MI355X holds it, and no host here holds it page-locked


@pytest.fixture(scope="module")
def lib():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cytvdn_amd import _lib
    return _lib


@pytest.fixture(autouse=True)
def tight_host_limit(monkeypatch):
    monkeypatch.setenv("TVDN_HOST_LIMIT", "1G")


def _big_args(lib, x, y, sums):
    a = lib.RunArgs(dtype=0, ndim=4, bc_mode=2, device=0, n_fista=2, n_plain=0)
    for i, s in enumerate(BIG):
        a.shape[i] = s
    a.data, a.recon_out, a.sums_out = x.ctypes.data, y.ctypes.data, sums.ctypes.data
    return a


def test_tvdn_plan_arithmetic(lib):
    L = lib.lib()
    out = lib.PlanOut()
    sh = lib.shape_arr((256, 256, 128, 128))
    assert L.tvdn_plan(0, 4, sh, 1, 1, 0, C.byref(out)) == 0
    assert out.arrays == 15 and out.bytes_per_slab >= 15 * 4 * 2 ** 30 and out.min_slabs >= 1
    # `fits` is 90 % of what is free NOW (earlier tests of this process may still hold HBM): check it as arithmetic
    assert out.fits == int(out.bytes_per_slab <= int(0.9 * out.free_bytes))
    big = lib.shape_arr(BIG)
    assert L.tvdn_plan(0, 4, big, 1, 1, 0, C.byref(out)) == 0 and out.fits == 0 and out.min_slabs > 8
    assert L.tvdn_plan(0, 4, sh, 0, 2, 0, C.byref(out)) == 0 and out.arrays == 11
    assert out.bytes_per_slab >= 11 * 130 * 256 * 128 * 128 * 4
    assert L.tvdn_plan(0, 4, sh, 1, 0, 0, C.byref(out)) == -1


def test_misfit_is_refused_before_anything_is_touched(lib):
    L = lib.lib()
    x = np.full(1 << 20, 7.0, np.float32)          # 4 MiB each, distinct; never read or written: every call is refused
    y = np.full(1 << 20, -3.0, np.float32)
    sums = np.zeros((2, 3))
    a = _big_args(lib, x, y, sums)
    # (1) never stream (0 / 0): refused with tvdn_plan's arithmetic in the message
    assert L.tvdn_run(C.byref(a)) == -2
    msg = L.tvdn_last_error().decode()
    assert "exceeds" in msg and "plan_run" in msg
    # (2) the host side as arithmetic: 11 cubes of 512 GiB (data, recon, 2 x 4 state arrays, and one more because two
    # arrays 4 MiB apart that claim 512 GiB each overlap) against a 1 GiB cap -> refusal, no device, no array touched
    need, avail = C.c_int64(), C.c_int64()
    assert L.tvdn_stream_host_need(C.byref(a), C.byref(need), C.byref(avail)) == -2
    assert need.value == 11 * 4 * int(np.prod(BIG)) and 0 < avail.value <= 1 << 30
    # (3) only now the streamed forms of the call itself: decide-yourself (-1 / -1) and explicit rows / k
    for rows, k in ((-1, -1), (2, 4)):
        a.stream_rows, a.stream_k = rows, k
        assert L.tvdn_run(C.byref(a)) == -2
        msg = L.tvdn_last_error().decode()
        # (with little HBM left free by earlier tests the refusal may already come from the ring arithmetic)
        assert ("host memory" in msg and "exceeds" in msg) or "fit the device" in msg or "bytes of HBM" in msg
    assert (x == 7.0).all() and (y == -3.0).all() and not sums.any()


def test_a_cube_the_cap_forbids_is_refused_even_when_it_would_fit(lib):
    """0.8 x TVDN_HOST_LIMIT is a hard ceiling for a streamed run: a 2 GiB-state cube that every box could hold is
    refused under the 1 GiB cap of this module (what protects the test-suite from its own mistakes)."""
    L = lib.lib()
    shape = (64, 64, 64, 64)                       # 64 MiB cube, 10 of them = 640 MiB < 0.8 GiB: allowed ...
    x = np.zeros((128,) + shape[1:], np.float32)   # (the arrays really have the 128 rows the second call claims)
    y = np.full_like(x, 5.0)
    sums = np.zeros((2, 3))
    a = lib.RunArgs(dtype=0, ndim=4, bc_mode=2, device=0, n_fista=2, n_plain=0, stream_rows=4, stream_k=2)
    for i, s in enumerate(shape):
        a.shape[i] = s
        a.clip[i], a.lambda_mu[i] = 32.0, 1 / 32
    a.data, a.recon_out, a.sums_out = x.ctypes.data, y.ctypes.data, sums.ctypes.data
    assert L.tvdn_stream_host_need(C.byref(a), None, None) == 0
    assert L.tvdn_run(C.byref(a)) == 0 and not y[:64].any() and (y[64:] == 5.0).all()
    a.shape[0] = 128                               # ... twice the rows: 1280 MiB > 0.8 GiB -> refused, y untouched
    y[:] = 5.0
    assert L.tvdn_stream_host_need(C.byref(a), None, None) == -2
    assert L.tvdn_run(C.byref(a)) == -2 and "host memory" in L.tvdn_last_error().decode()
    assert (y == 5.0).all()
