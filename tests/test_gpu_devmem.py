"""The state's device memory composed from physical granules (csrc/tvdn_devmem.hip, tvdn_mem_alloc): what a
run computes must not depend on what its state is made of -- the bits are the oracle's (cyTVDN/cyTVDN.py:148-242) on granules, on
a plain hipMalloc block and on a caller's workspace alike -- and the run says which it was (tvdn_run_stats.state_mem)."""
import ctypes as C

import numpy as np
import pytest

from golden_util import bits_equal

pytestmark = pytest.mark.gpu


def test_mem_alloc_kinds_and_torch_view():
    import torch
    from cytvdn_amd import _lib
    big = _lib.DeviceBlock(3 << 30, 0)                      # >= 2 GiB: granules (1 GiB each, the last one cut to size)
    small = _lib.DeviceBlock(1 << 20, 0)
    assert small.kind == _lib.MEM_PLAIN
    assert big.kind == _lib.MEM_GRANULES, "this runtime refused hipMemCreate / hipMemMap: the product would fall back to plain blocks"
    t = big.tensor(torch.float32)
    assert t.data_ptr() == big.ptr and t.numel() == (3 << 30) // 4 and t.device.index == 0
    t.fill_(1.5)                                            # every granule is mapped read-write: touch all of it
    assert float(t[::4096].sum()) == 1.5 * t[::4096].numel()
    t[-1] = 7.0
    assert float(t[-1]) == 7.0
    del t
    big.free()
    small.free()
    big.free()                                              # idempotent


def test_odd_sizes_round_trip(monkeypatch):
    """Blocks that are not a multiple of the granule are rounded up to whole granules of ONE size (handles of different sizes
    in one reservation break the runtime's own copies on ROCm 7.2: csrc/tvdn_devmem.hip); every byte asked for is there."""
    import torch
    from cytvdn_amd import _lib
    monkeypatch.setenv("TVDN_VMM_MIN_MIB", "1")
    monkeypatch.setenv("TVDN_GRANULE_MIB", "8")
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info(0)[0]
    for nbytes in (8 << 20, (8 << 20) + 16, (40 << 20) - 4096, (17 << 20) + 4):
        b = _lib.DeviceBlock(nbytes, 0)
        assert b.kind == _lib.MEM_GRANULES
        t = b.tensor(torch.uint8)
        t.fill_(3)
        assert int(t[-1]) == 3 and int(t[0]) == 3 and int(t[nbytes // 2]) == 3 and t.numel() == nbytes
        assert int(t.sum(dtype=torch.int64)) == 3 * nbytes
        del t
        b.free()
    import time
    torch.cuda.empty_cache()            # (the sums' temporaries sit in torch's cache)
    for _ in range(100):                # released granules come back once the driver has cleared them: a moment later
        if free0 - torch.cuda.mem_get_info(0)[0] < (64 << 20):
            break
        time.sleep(0.1)
    # nothing leaked (one-sided: what EARLIER tests released may still have been on its way back when free0 was read -- seen once:
    # 190 MB more free after the test than before it)
    assert free0 - torch.cuda.mem_get_info(0)[0] < (64 << 20)


@pytest.mark.parametrize("granule_mib,nbytes", [(8, 40 << 20), (1024, (3 << 30) + 4096)])
def test_runtime_copies_across_granule_borders(monkeypatch, granule_mib, nbytes):
    """What tvdn_run does to its state with the RUNTIME's own calls -- fills, host-to-device rows, device-to-device rows,
    device-to-host rows -- must land where a kernel sees it, also where a transfer straddles two granules."""
    import torch
    from cytvdn_amd import _lib
    monkeypatch.setenv("TVDN_VMM_MIN_MIB", "1")
    monkeypatch.setenv("TVDN_GRANULE_MIB", str(granule_mib))
    b = _lib.DeviceBlock(nbytes, 0)
    assert b.kind == _lib.MEM_GRANULES
    t = b.tensor(torch.uint8)
    G = granule_mib << 20
    rng = np.random.default_rng(5)
    t.zero_()                                                                    # hipMemsetAsync over every granule
    assert int(t.sum(dtype=torch.int64)) == 0
    span = 6 << 20
    for border in range(G, nbytes - span, G):
        lo = border - span // 2 + 123                                             # odd offsets, straddling the border
        src = rng.integers(0, 256, span, dtype=np.uint8)
        t[lo - 1], t[lo + span] = 201, 202                                        # sentinels either side
        _lib.copy_to_device(src, t[lo:lo + span])                                # the library's pinned multi-lane upload
        assert np.array_equal(t[lo:lo + span].cpu().numpy(), src)                # torch's device-to-host copy
        assert np.array_equal(_lib.copy_to_host(t[lo:lo + span], np.uint8), src)  # the library's download
        dst = 4096 + 77 if lo > 2 * span else nbytes - span - 4096 - 77          # device to device, away from the source
        t[dst:dst + span].copy_(t[lo:lo + span])
        assert bool((t[dst:dst + span] == torch.from_numpy(src).cuda()).all())
        t[lo:lo + span].zero_()                                                  # a fill that straddles the border
        assert int(t[lo:lo + span].sum(dtype=torch.int64)) == 0
        assert int(t[lo - 1]) == 201 and int(t[lo + span]) == 202
        assert bool((t[dst:dst + span] == torch.from_numpy(src).cuda()).all())
    del t
    b.free()


@pytest.mark.parametrize("shape,dtype,its,fista,bc", [
    ((9, 5, 8, 16), np.float32, 7, True, 2), ((12, 6, 16), np.float64, [3, 2], True, 2), ((7, 3, 4, 8), np.float32, 5, False, 0),
    ((40, 12, 32, 64), np.float32, 4, True, 2),
])
def test_runs_on_granules_give_the_oracles_bits(oracle, monkeypatch, shape, dtype, its, fista, bc):
    """denoise3D/4D with the state on granules (forced down to small cubes), on a plain block, and through the Python loop."""
    import cytvdn_amd as tv
    from cytvdn_amd import synth
    dt = np.dtype(dtype)
    nd = len(shape)
    x = synth.cube(shape, seed=31, dtype=dt) + dt.type(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6][:nd], dt)
    fn = tv.denoise4D if nd == 4 else tv.denoise3D
    ref = oracle.denoise(x, mu, its, fista, BC_mode=bc)
    monkeypatch.setenv("TVDN_GRANULE_MIB", "2")
    for vmm, min_mib, loop in (("1", "1", "run"), ("0", "1", "run"), ("1", "1", "python"), ("1", "2048", "run")):
        monkeypatch.setenv("TVDN_VMM", vmm)
        monkeypatch.setenv("TVDN_VMM_MIN_MIB", min_mib)
        monkeypatch.setenv("TVDN_LOOP", loop)
        recon, bn, dl = fn(x, mu, its, FISTA=fista, BC_mode=bc, quiet=True)
        assert bits_equal(recon, ref["recon"]), (vmm, min_mib, loop)
        np.testing.assert_allclose(bn, ref["b_norm64"], rtol=1e-5)
    from cytvdn_amd import _lib
    _lib.lib().tvdn_release_cache()


def _run_args(x, mu, n_f, n_p, recon, sums, stats):
    from cytvdn_amd import _lib
    dt, nd = x.dtype, x.ndim
    lam = mu / dt.type(32.0 if nd == 4 else 16.0)
    a = _lib.RunArgs(dtype=_lib.dtype_code(dt), ndim=nd, bc_mode=2, device=0, n_fista=n_f, n_plain=n_p)
    for i, s_ in enumerate(x.shape):
        a.shape[i] = s_
    for q in range(nd):
        a.clip[q] = float((1.0 / lam)[q])
        a.lambda_mu[q] = float((lam / mu).astype(dt)[q])
    a.data, a.recon_out, a.sums_out = x.ctypes.data, recon.ctypes.data, sums.ctypes.data
    a.stats = C.addressof(stats)
    return a


def test_tvdn_run_says_what_its_state_is_made_of(oracle, monkeypatch):
    """tvdn_run_stats.state_mem: granules / plain / the caller's workspace; no audition on granules (there is nothing to choose
    between), the audition of plain blocks as before; the kept block counts as free for the planner."""
    import torch
    from cytvdn_amd import _lib, planner, synth
    shape, n_f = (10, 6, 16, 32), 5
    x = synth.cube(shape, seed=37, dtype=np.float32) + np.float32(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6], np.float32)
    ref = oracle.denoise(x, mu, n_f, True)
    monkeypatch.setenv("TVDN_GRANULE_MIB", "2")
    monkeypatch.setenv("TVDN_VMM_MIN_MIB", "1")
    seen = []
    for vmm, aud in (("1", None), ("0", None), ("0", "3"), ("1", "2")):
        _lib.lib().tvdn_release_cache()
        monkeypatch.setenv("TVDN_VMM", vmm)
        if aud is None:
            monkeypatch.delenv("TVDN_AUDITION", raising=False)
        else:
            monkeypatch.setenv("TVDN_AUDITION", aud)
        recon, sums, stats = np.empty_like(x), np.zeros((n_f, 3)), _lib.RunStats()
        a = _run_args(x, mu, n_f, 0, recon, sums, stats)
        _lib.check(_lib.lib().tvdn_run(C.byref(a)))
        assert bits_equal(recon, ref["recon"]), (vmm, aud)
        seen.append((stats.state_mem, stats.audition_n))
    assert seen[0] == (_lib.MEM_GRANULES, 0) and seen[1] == (_lib.MEM_PLAIN, 0)
    assert seen[2] == (_lib.MEM_PLAIN, 3) and seen[3] == (_lib.MEM_GRANULES, 2)     # TVDN_AUDITION insists: obeyed on either kind
    # the block the last run kept: used as far as the driver says, free as far as planning goes
    kept = _lib.state_kept_bytes(0)
    assert kept > 0
    free = torch.cuda.mem_get_info(0)[0]
    monkeypatch.delenv("TVDN_HBM_LIMIT", raising=False)
    assert planner.hbm_available(0) >= free + kept
    # a caller's workspace
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda:0")
    recon, sums, stats = np.empty_like(x), np.zeros((n_f, 3)), _lib.RunStats()
    a = _run_args(x, mu, n_f, 0, recon, sums, stats)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    _lib.check(_lib.lib().tvdn_run(C.byref(a)))
    assert bits_equal(recon, ref["recon"]) and stats.state_mem == _lib.MEM_CALLER
    _lib.lib().tvdn_release_cache()
    assert _lib.state_kept_bytes(0) == 0


def test_streamed_run_on_granules(oracle, monkeypatch):
    """The streamed engine's device block (rings, boxes, kept rows) comes from the same allocator."""
    from cytvdn_amd import _lib, synth
    shape, n_f = (24, 6, 16, 32), 6
    x = synth.cube(shape, seed=41, dtype=np.float32) + np.float32(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6], np.float32)
    ref = oracle.denoise(x, mu, n_f, True)
    monkeypatch.setenv("TVDN_GRANULE_MIB", "2")
    monkeypatch.setenv("TVDN_VMM_MIN_MIB", "1")
    for vmm, want in (("1", _lib.MEM_GRANULES), ("0", _lib.MEM_PLAIN)):
        _lib.lib().tvdn_release_cache()
        monkeypatch.setenv("TVDN_VMM", vmm)
        recon, sums, stats = np.empty_like(x), np.zeros((n_f, 3)), _lib.RunStats()
        a = _run_args(x, mu, n_f, 0, recon, sums, stats)
        a.stream_rows, a.stream_k, a.stream_resident = 2, 3, -1
        _lib.check(_lib.lib().tvdn_run(C.byref(a)))
        assert bits_equal(recon, ref["recon"]), vmm
        assert stats.engine == 1 and stats.state_mem == want
    _lib.lib().tvdn_release_cache()
    _lib.lib().tvdn_wait_background()


def test_a_kept_block_changes_size_with_the_granules_it_has(oracle, monkeypatch):
    """Runs of different sizes one after the other in one process: the kept block of the last run is not freed and allocated anew
    but re-dealt at the new size from the granules it has (tvdn_devmem.hip dev_resize: missing ones created, surplus ones given
    back, a new random order, fresh translations) -- growing, shrinking, growing again, resident and streamed, each the oracle's bits."""
    import torch
    from cytvdn_amd import _lib, synth
    monkeypatch.setenv("TVDN_GRANULE_MIB", "2")        # small cubes on many granules
    monkeypatch.setenv("TVDN_VMM_MIN_MIB", "1")
    monkeypatch.setenv("TVDN_VMM", "1")
    _lib.lib().tvdn_release_cache()
    mu = np.array([1.0, 0.8, 0.5, 0.6], np.float32)
    free0 = torch.cuda.mem_get_info(0)[0]
    kept_before = None
    for rows, stream in ((10, None), (22, None), (6, None), (30, (3, 3)), (12, None), (40, None)):
        shape = (rows, 6, 16, 32)
        x = synth.cube(shape, seed=rows, dtype=np.float32) + np.float32(0.25)
        ref = oracle.denoise(x, mu, 5, True)
        recon, sums, stats = np.empty_like(x), np.zeros((5, 3)), _lib.RunStats()
        a = _run_args(x, mu, 5, 0, recon, sums, stats)
        if stream:
            a.stream_rows, a.stream_k, a.stream_resident = stream[0], stream[1], 0
        _lib.check(_lib.lib().tvdn_run(C.byref(a)))
        assert bits_equal(recon, ref["recon"]), (rows, stream)
        assert stats.state_mem == _lib.MEM_GRANULES
        kept = _lib.state_kept_bytes(0)
        assert kept > 0 and kept != kept_before           # a block of THIS run's size is what is kept now
        kept_before = kept
    _lib.lib().tvdn_release_cache()
    assert _lib.state_kept_bytes(0) == 0
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(0)[0] >= free0 - (64 << 20)      # every granule went back


def test_a_short_first_run_takes_few_spare_granules_and_a_longer_one_tops_the_block_up(oracle, monkeypatch):
    """Round 6: tvdn_run gives the spare granules 5 % of its expected sweep time and no longer a floor of
    0.25 s -- the first 50-iteration call of a process is not worth a quarter of a second of pool -- and the kept block, chosen from
    a short pool, is re-drawn from its own and new granules by the first run that can afford it (dev_upgrade in state_acquire):
    `first_call` tells the two apart, the bits are the oracle's either way, and nothing leaks."""
    import torch
    from cytvdn_amd import _lib, synth
    monkeypatch.setenv("TVDN_GRANULE_MIB", "2")
    monkeypatch.setenv("TVDN_VMM_MIN_MIB", "1")
    monkeypatch.setenv("TVDN_VMM", "1")
    _lib.lib().tvdn_release_cache()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info(0)[0]
    shape, n_f = (30, 6, 16, 32), 5
    x = synth.cube(shape, seed=47, dtype=np.float32) + np.float32(0.25)
    mu = np.array([1.0, 0.8, 0.5, 0.6], np.float32)
    ref = oracle.denoise(x, mu, n_f, True)
    seen = []
    for budget in ("0", "0", "0.5", "0.5"):
        monkeypatch.setenv("TVDN_SPREAD_S", budget)       # (what the run's length would otherwise decide)
        recon, sums, stats = np.empty_like(x), np.zeros((n_f, 3)), _lib.RunStats()
        a = _run_args(x, mu, n_f, 0, recon, sums, stats)
        _lib.check(_lib.lib().tvdn_run(C.byref(a)))
        assert bits_equal(recon, ref["recon"]), budget
        st = _lib.mem_status(0)
        seen.append((stats.first_call, st["last_granules"], st["last_pool"]))
    need = seen[0][1]
    assert seen[0] == (1, need, need) and seen[1] == (0, need, need)          # no spare granules, and the kept block taken as it is
    assert seen[2][0] == 0 and seen[2][1] == need and seen[2][2] >= 2 * need    # topped up by the run that could afford it ...
    assert seen[3] == seen[2]                                                   # ... once
    assert _lib.mem_status(0)["faults"] == 0
    _lib.lib().tvdn_release_cache()
    torch.cuda.synchronize()
    import time
    for _ in range(100):
        if torch.cuda.mem_get_info(0)[0] >= free0 - (64 << 20):
            break
        time.sleep(0.05)
    assert torch.cuda.mem_get_info(0)[0] >= free0 - (64 << 20)
