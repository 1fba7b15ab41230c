"""Host-side contract of the drop-in boundary, checked WITHOUT a GPU:
 * the C-ABI library loads and exports every symbol include/tvdn.h declares;
 * the Python surface has the reference's names, signatures and defaults (cyTVDN/cyTVDN.py:19-31, :250-260);
 * argument errors surface as the reference's exception types BEFORE any device work;
 * with no GPU the product raises (no CPU fallback, and nothing under cytvdn_amd imports oracle/).
"""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest
import torch

import cytvdn_amd as tv
from cytvdn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_GPU = not torch.cuda.is_available()


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "tvdn.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tvdn_[a-z_0-9]+)\s*\(", hdr))
    assert {"tvdn_iterate_fused", "tvdn_accumulator_update", "tvdn_datacube_update", "tvdn_sum_square_error",
            "tvdn_ctx_create", "tvdn_synth_fill"} <= declared
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/tvdn.h but not exported"
    assert declared == set(_lib.EXPORTS)
    assert _lib.lib().tvdn_abi_version() == 9


def test_iter_args_struct_matches_header_layout():
    # 2 int32 + 4 int64 + 2 int64 + 4 int32 + 2 double + 8 doubles + 3 ptr + 20 ptr + 2 int64 + 2 int32 + 1 ptr (ABI v2) + 2 int64 (ABI v3: row rings)
    # + 5 int64 (ABI 8: ring sizes of their own for recon_in, the state of this level, of the level before, recon_out, the outputs)
    assert ctypes.sizeof(_lib.IterArgs) == 8 + 32 + 16 + 16 + 16 + 64 + 24 + 160 + 16 + 8 + 8 + 16 + 40
    assert _lib.IterArgs.recon_in_ring_rows.offset == _lib.IterArgs.ring_rows.offset + 16 and _lib.IterArgs.out_ring_rows.offset == _lib.IterArgs.ring_rows.offset + 48
    assert _lib.IterArgs.shape.offset == 8 and _lib.IterArgs.tk.offset == 72 and _lib.IterArgs.orig.offset == 152
    assert _lib.IterArgs.dprev_in.offset == 304 and _lib.IterArgs.sweep_lo.offset == 336


def test_public_names_and_signatures():
    for n in ("denoise4D", "denoise3D", "check_memory", "accumulator_update_4D", "accumulator_update_4D_FISTA",
              "accumulator_update_3D", "accumulator_update_3D_FISTA", "datacube_update_4D", "datacube_update_3D",
              "sum_square_error_4D", "sum_square_error_3D", "iso_accumulator_update_4D",
              "iso_accumulator_update_4D_FISTA"):
        assert callable(getattr(tv, n))
    p4 = inspect.signature(tv.denoise4D).parameters
    assert list(p4)[:11] == ["datacube", "mu", "iterations", "FISTA", "stopping_relative_change", "isotropic_R",
                             "isotropic_Q", "reference_data", "BC_mode", "lam", "quiet"]
    assert (p4["iterations"].default, p4["FISTA"].default, p4["BC_mode"].default, p4["quiet"].default) == (10, True, 2, False)
    p3 = inspect.signature(tv.denoise3D).parameters
    assert list(p3)[:9] == ["datacube", "mu", "iterations", "stopping_relative_change", "BC_mode", "FISTA",
                            "reference_data", "lam", "quiet"]
    assert (p3["iterations"].default, p3["FISTA"].default, p3["BC_mode"].default) == (7500, False, 2)
    pa = inspect.signature(tv.accumulator_update_4D_FISTA).parameters
    assert list(pa) == ["a", "b", "d", "tk", "ax", "clip", "BC_mode"] and pa["BC_mode"].default == 2
    pd = inspect.signature(tv.datacube_update_3D).parameters
    assert list(pd) == ["orig", "recon", "b1", "b2", "b3", "lambda_mu", "BC_mode"]


def test_driver_assertions_match_reference():
    x = np.zeros((2, 3, 4, 4), np.float32)
    mu = np.ones(4, np.float32)
    with pytest.raises(AssertionError, match="floating point"):
        tv.denoise4D(x.astype(np.int32), mu, 1, quiet=True)
    with pytest.raises(AssertionError, match="Lambda must have same dtype"):
        tv.denoise4D(x, mu, 1, lam=np.ones(4, np.float64) / 32, quiet=True)
    with pytest.raises(AssertionError, match="Mu must have same dtype"):
        tv.denoise4D(x, np.ones(4, np.float64), 1, lam=mu / 32, quiet=True)
    with pytest.raises(AssertionError, match="C-contiguous"):
        tv.denoise4D(np.asfortranarray(x), mu, 1, quiet=True)
    with pytest.raises(NotImplementedError):
        tv.denoise4D(x, mu, 1, isotropic_R=True, quiet=True)
    with pytest.raises(NotImplementedError):
        tv.denoise4D(x, mu, 1, BC_mode=1, quiet=True)
    y = np.zeros((3, 4, 4), np.float64)
    with pytest.raises(AssertionError, match="Lambda must have same dtype"):
        tv.denoise3D(y, np.ones(3, np.float32), 1, quiet=True)     # wrong-dtype mu trips the lam assert (cyTVDN.py:297)
    with pytest.raises(AssertionError, match="Parameters must satisfy"):
        tv.denoise3D(y, np.ones(3), 1, lam=np.ones(3) / 8, quiet=True)
    with pytest.raises(TypeError, match="No matching signature"):
        tv.denoise4D(y, np.ones(4), 1, quiet=True)
    with pytest.raises(TypeError, match="No matching signature"):
        tv.denoise3D(x, np.ones(3, np.float32), 1, quiet=True)


def test_kernel_level_argument_errors():
    a = np.zeros((2, 3, 4, 4), np.float32)
    with pytest.raises(TypeError, match="No matching signature"):
        tv.accumulator_update_4D(a[0], a[0], 0, 1.0)                       # wrong rank
    with pytest.raises(TypeError, match="No matching signature"):
        tv.accumulator_update_4D(a.astype(np.int32), a.astype(np.int32), 0, 1.0)
    with pytest.raises(ValueError, match="Buffer dtype mismatch, expected 'float' but got 'double'"):
        tv.accumulator_update_4D(a, a.astype(np.float64), 0, 1.0)
    ro = a.copy()
    ro.flags.writeable = False
    with pytest.raises(ValueError, match="read-only"):
        tv.accumulator_update_4D(ro, a.copy(), 0, 1.0)
    with pytest.raises(ValueError, match="ax"):
        tv.accumulator_update_4D(a, a.copy(), 4, 1.0)
    with pytest.raises(ValueError, match="BC_mode"):
        tv.accumulator_update_4D(a, a.copy(), 0, 1.0, BC_mode=7)
    with pytest.raises(ValueError, match="shapes disagree"):
        tv.accumulator_update_4D(a, np.zeros((2, 3, 4, 5), np.float32), 0, 1.0)
    with pytest.raises(NotImplementedError):
        tv.datacube_update_4D(a, a.copy(), a, a, a, a, np.ones(4, np.float32), BC_mode=1)
    with pytest.raises(ValueError, match="Buffer dtype mismatch"):
        tv.datacube_update_4D(a, a.copy(), a, a, a, a, np.ones(4, np.float64))
    with pytest.raises(NotImplementedError):
        tv.iso_accumulator_update_4D(a, a, a, 0, 1, 1.0)


@pytest.mark.skipif(not NO_GPU, reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback():
    x = np.ones((2, 3, 4, 4), np.float32)
    with pytest.raises(_lib.TvdnError, match="no CPU fallback"):
        tv.denoise4D(x, np.ones(4, np.float32), 2, quiet=True)
    with pytest.raises(_lib.TvdnError, match="no CPU fallback"):
        tv.sum_square_error_4D(x, x)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "cytvdn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libtvdn_oracle" not in src and "oracle/_ref" not in src, f


def test_hbm_plan_counts_arrays():
    from cytvdn_amd.engine import hbm_plan
    assert hbm_plan((256, 256, 128, 128), np.float32, True)["arrays"] == 15
    assert hbm_plan((256, 256, 128, 128), np.float32, True)["bytes"] == 15 * 4 * 2 ** 30
    assert hbm_plan((256, 256, 128, 128), np.float64, False)["arrays"] == 11
    assert hbm_plan((128, 128, 512), np.float32, True)["arrays"] == 12


def test_empty_cube_needs_no_gpu():
    """A zero-size axis: upstream's loops fall through (norms 0, delta_recon 0/0 = NaN, recon an empty copy)."""
    x = np.zeros((3, 0, 4, 4), np.float32)
    recon, bn, dl = tv.denoise4D(x, np.ones(4, np.float32), 3, quiet=True)
    assert recon.shape == x.shape and recon is not x
    assert bn.dtype == np.float32 and np.array_equal(bn, np.zeros(3, np.float32)) and np.isnan(dl).all()
    y = np.zeros((0, 5, 6), np.float64)
    out = tv.denoise3D(y, np.ones(3), [2, 1], reference_data=y.copy(), quiet=True)
    assert len(out) == 4 and out[3].shape == (4,) and out[0].shape == y.shape


def test_struct_layouts_match_the_c_header(tmp_path):
    """The ctypes mirrors against the real thing: a C program built with gcc from include/tvdn.h prints size and field
    offsets of every struct that crosses the boundary."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "layout.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "tvdn.h"\n'
                   '#define P(s, f) printf(#s "." #f " %zu\\n", offsetof(s, f))\n'
                   'int main(void) {\n'
                   '  printf("tvdn_iter_args %zu\\n", sizeof(tvdn_iter_args)); P(tvdn_iter_args, wrap_recon); P(tvdn_iter_args, ring_rows);\n'
                   '  P(tvdn_iter_args, orig_ring_rows); P(tvdn_iter_args, accumulate);\n'
                   '  printf("tvdn_many_args %zu\\n", sizeof(tvdn_many_args)); P(tvdn_many_args, recon); P(tvdn_many_args, S); P(tvdn_many_args, tk_prev);\n'
                   '  printf("tvdn_run_args %zu\\n", sizeof(tvdn_run_args)); P(tvdn_run_args, stop); P(tvdn_run_args, data); P(tvdn_run_args, devices);\n'
                   '  P(tvdn_run_args, stream_rows); P(tvdn_run_args, stream_k); P(tvdn_run_args, phase_iters); P(tvdn_run_args, progress_user); P(tvdn_run_args, workspace_bytes); P(tvdn_run_args, n_devices); P(tvdn_run_args, stats); P(tvdn_run_args, stream_resident); P(tvdn_run_args, slab);\n'
                   '  printf("tvdn_slab_io %zu\\n", sizeof(tvdn_slab_io)); P(tvdn_slab_io, row0); P(tvdn_slab_io, first_row_nonfinite); P(tvdn_slab_io, exchange); P(tvdn_slab_io, relay_row0); P(tvdn_slab_io, user);\n'
                   '  printf("tvdn_run_stats %zu\\n", sizeof(tvdn_run_stats)); P(tvdn_run_stats, resident_rows); P(tvdn_run_stats, d2h_bytes); P(tvdn_run_stats, total_s);\n'
                   '  P(tvdn_run_stats, audition_kept); P(tvdn_run_stats, audition_ms); P(tvdn_run_stats, first_pass_s); P(tvdn_run_stats, first_pass_iters);\n'
                   '  printf("tvdn_plan_out %zu\\n", sizeof(tvdn_plan_out)); P(tvdn_plan_out, fits); P(tvdn_plan_out, min_slabs);\n'
                   '  return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    got = dict(line.rsplit(" ", 1) for line in subprocess.check_output([str(exe)], text=True).splitlines())
    mirrors = {"tvdn_iter_args": _lib.IterArgs, "tvdn_many_args": _lib.ManyArgs, "tvdn_run_args": _lib.RunArgs,
               "tvdn_plan_out": _lib.PlanOut, "tvdn_run_stats": _lib.RunStats, "tvdn_slab_io": _lib.SlabIO}
    for key, val in got.items():
        name, _, field = key.partition(".")
        want = getattr(mirrors[name], field).offset if field else ctypes.sizeof(mirrors[name])
        assert int(val) == want, (key, val, want)


@pytest.mark.skipif(not NO_GPU, reason="only meaningful on a box without a GPU")
def test_tvdn_run_checks_its_arguments_before_it_looks_for_a_device():
    a = _lib.RunArgs(dtype=0, ndim=4, bc_mode=2, n_fista=1, stream_rows=4, stream_k=0)
    for i, s in enumerate((4, 3, 4, 8)):
        a.shape[i] = s
    x = np.zeros((4, 3, 4, 8), np.float32)
    sums = np.zeros((1, 3))
    a.data, a.recon_out, a.sums_out = x.ctypes.data, x.ctypes.data, sums.ctypes.data
    L = _lib.lib()
    assert L.tvdn_run(ctypes.byref(a)) == -1 and "must both be 0" in L.tvdn_last_error().decode()
    a.stream_rows = a.stream_k = -1
    assert L.tvdn_run(ctypes.byref(a)) == -4 and "no CPU fallback" in L.tvdn_last_error().decode()


def test_progress_bars_follow_the_library_reports(monkeypatch):
    """driver._ProgressBars turns tvdn_run's slot counts into upstream's two bars (cyTVDN.py:148-151, :196-199): per
    iteration, per pipelined phase (one report may end the FISTA phase and start the unaccelerated one), and with a
    stopping rule, where a phase that is left early keeps its count."""
    from cytvdn_amd import driver
    closed = []

    class Bar:
        def __init__(self, total, desc):
            self.total, self.desc, self.n = total, desc, 0

        def update(self, k):
            assert k >= 0
            self.n += k

        def close(self):
            closed.append((self.desc.split()[0], self.n, self.total))

    monkeypatch.setattr(driver, "_tqdm", Bar)

    def play(n_f, n_p, reports, **kw):
        closed.clear()
        b = driver._ProgressBars(n_f, n_p, False, **kw)
        assert b.active
        for r in reports:
            b.update(r)
        b.close()
        return list(closed)

    assert play(6, 4, range(1, 11)) == [("FISTA", 6, 6), ("Unaccelerated", 4, 4)]
    assert play(6, 4, [3, 4, 10]) == [("FISTA", 6, 6), ("Unaccelerated", 4, 4)]
    assert play(6, 4, [10]) == [("FISTA", 6, 6), ("Unaccelerated", 4, 4)]
    assert play(6, 4, [1, 7], may_stop=True) == [("FISTA", 1, 6), ("Unaccelerated", 1, 4)]
    assert play(0, 4, [2, 4]) == [("Unaccelerated", 4, 4)]
    assert play(5, 0, [5]) == [("FISTA", 5, 5)]
    assert not driver._ProgressBars(5, 0, True).active


def test_integration_md_structs_are_the_real_ones():
    """The ctypes stubs INTEGRATION.md shows a cyTVDN maintainer (IterArgs, RunArgs) are executed and compared with the
    tested binding field by field: name, ctypes type, offset."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    for name, real in (("IterArgs", _lib.IterArgs), ("RunArgs", _lib.RunArgs)):
        m = re.search(r"class %s\(C\.Structure\):.*?\n(\s+_fields_ = \[.*?\)\])" % name, text, re.S)
        assert m, name
        ns = {"C": ctypes}
        exec("class S(C.Structure):\n" + m.group(1), ns)
        doc = ns["S"]
        assert ctypes.sizeof(doc) == ctypes.sizeof(real), name
        assert [f[0] for f in doc._fields_] == [f[0] for f in real._fields_], name
        for f in doc._fields_:
            assert getattr(doc, f[0]).offset == getattr(real, f[0]).offset, (name, f[0])
            assert getattr(doc, f[0]).size == getattr(real, f[0]).size, (name, f[0])
    assert "tvdn_abi_version() == %d" % _lib.lib().tvdn_abi_version() in text
