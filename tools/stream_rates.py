#!/usr/bin/env python3
"""Streamed tvdn_run on synthetic cubes, one JSON line per run (bench.api_streamed): passes-only rate, PCIe GB/s, set-up and
whole-call seconds, resident rows.

    python tools/stream_rates.py SHAPE ROWS K ITERS [RESIDENT]     e.g.  64x1024x256x256 2 40 80 0      (-1 -1: the library's choice)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    shape = tuple(int(v) for v in sys.argv[1].split("x"))
    rows, k, iters = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    resident = int(sys.argv[5]) if len(sys.argv) > 5 else -1     # rows kept in HBM: -1 = as many as the plan says, 0 = none
    os.environ.setdefault("OMP_NUM_THREADS", str(bench.host_cores()))
    e = bench.api_streamed(shape, rows, k, iters, f"streamed tvdn_run {sys.argv[1]} rows {rows} k {k}", None, 0,
                           force_stream=True, resident=resident)
    print(json.dumps(e), flush=True)


if __name__ == "__main__":
    main()
