#!/usr/bin/env python3
"""Does the order of streamed runs inside one process matter?  (bench.py: hybrid plan first, then every row streamed.)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
os.environ["TVDN_STREAM_TIMING"] = "1"
half = (64, 1024, 256, 256)
x = bench.synth_host(half, 0)
order = sys.argv[1] if len(sys.argv) > 1 else "hs"
for c in order:
    if c == "h":
        e = bench.api_streamed(half, -1, -1, 80, "hybrid", x, 0, force_stream=True, resident=-1)
    else:
        e = bench.api_streamed(half, -1, -1, -2, "streamed", x, 0, force_stream=True, resident=0)
    print(json.dumps({k: e.get(k) for k in ("value", "stream_k", "resident_rows", "passes", "passes_s", "setup_s", "whole_call_s", "h2d_GBps", "d2h_GBps")}), flush=True)
