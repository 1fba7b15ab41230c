"""Measurement aid: what a hybrid streamed run (rows kept in HBM + rows streamed) of half a config-5 rank slab costs by depth.
   python tools/ubench/resident_rows_probe.py fresh|after|shapes [k ...]    (profiles/r05_hybrid_depths.jsonl)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from cytvdn_amd import _lib
import cytvdn_amd as tv
mode = sys.argv[1]
half = (int(os.environ.get("PROBE_ROWS", "64")), 1024, 256, 256)
def free(tag):
    f, t = torch.cuda.mem_get_info(0)
    print(tag, round(f/2**30, 2), "kept", round(_lib.state_kept_bytes(0)/2**30, 2), flush=True)
free("start")
if mode == "after":
    x2 = bench.synth_host((256,256,128,128), 0)
    free("after synth")
    tv.denoise4D(x2, np.array([1,1,.5,.5], np.float32), 20, quiet=True, device=0)
    free("after denoise4D")
    _lib.lib().tvdn_release_cache(); torch.cuda.empty_cache()
    free("after release")
x = bench.synth_host(half, 0)
free("after synth half")
e = bench.api_streamed(half, -1, -1, 24, "probe", x, 0, force_stream=True, resident=-1)
print({k: v for k, v in e.items() if k != "config"}, flush=True)
free("end")
if mode == "shapes":      # 80 iterations at explicit depths, the rows kept being what fits beside the rings of each
    import json
    if not os.environ.get("PROBE_SKIP_PLAN"):
        e = bench.api_streamed(half, -1, -1, 80, "the library's own plan for 80 iterations", x, 0, force_stream=True, resident=-1)
        print(json.dumps({kk: v for kk, v in e.items() if kk != "config"}), flush=True)
    for spec in sys.argv[2:] or ("8", "10", "12", "14", "16", "20"):     # K or R:K
        r, k = (int(v) for v in spec.split(":")) if ":" in spec else (2, int(spec))
        e = bench.api_streamed(half, r, k, int(os.environ.get("PROBE_ITERS", "80")), f"{r}-row chunks, k = {k}", x, 0, force_stream=False,
                               resident=int(os.environ.get("PROBE_RESIDENT", "-1")))
        print(json.dumps({kk: v for kk, v in e.items() if kk in ("value", "iterations", "stream_rows", "stream_k", "resident_rows", "passes", "passes_s", "setup_s", "h2d_GBps", "d2h_GBps")}), flush=True)
