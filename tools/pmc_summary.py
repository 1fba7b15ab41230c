#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter CSVs into the rows that matter (tvdn kernels + the calibration copy) and, given a
FETCH_SIZE pass and a WRITE_SIZE pass of the same command, into per-launch HBM-side traffic:

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_x  [--traffic-json profiles/traffic.json]

Writes <prefix>_pmc_fetch_size.csv / <prefix>_pmc_write_size.csv (kernel, grid, counter, value, timestamps) and prints
one line per (kernel, grid): launches, mean FETCH_SIZE, mean WRITE_SIZE (KiB) and traffic = (2*FETCH + WRITE) * 1024 B
(gfx950: FETCH_SIZE reports half of a wide coalesced read, MI355X_MICROARCH.md section HBM; checked on every run against
the device-to-device copy `be.recon[cur].copy_(be.orig)` the benchmark itself makes)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def rows_of(d):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"]
                if "tvdn::" in name or "copyBuffer" in name:
                    out.append((name, int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"]),
                                int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    out.sort(key=lambda t: t[4])
    return out


def main():
    fetch_dir, write_dir, prefix = sys.argv[1:4]
    tj = sys.argv[sys.argv.index("--traffic-json") + 1] if "--traffic-json" in sys.argv else None
    acc = defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
    for d, cname in ((fetch_dir, "fetch_size"), (write_dir, "write_size")):
        rows = rows_of(d)
        with open(f"{prefix}_pmc_{cname}.csv", "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
            w.writerows(rows)
        for name, grid, counter, val, _, _ in rows:
            acc[(name, grid)][counter].append(val)
    summary = {}
    for (name, grid), c in sorted(acc.items()):
        f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) if c["FETCH_SIZE"] else None
        w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) if c["WRITE_SIZE"] else None
        tr = (2 * f + w) * 1024 if f is not None and w is not None else None
        short = name.replace("void ", "").split("(")[0]
        summary[f"{short}|grid={grid}"] = dict(launches=max(len(c["FETCH_SIZE"]), len(c["WRITE_SIZE"])),
                                               fetch_size_kib=f, write_size_kib=w, traffic_bytes=tr)
        print(f"{short:70s} grid={grid:>11d} n={len(c['FETCH_SIZE']):3d}/{len(c['WRITE_SIZE']):3d} "
              f"FETCH={f if f is None else round(f, 1)} KiB WRITE={w if w is None else round(w, 1)} KiB "
              f"traffic={tr if tr is None else round(tr / 1e9, 3)} GB")
    with open(f"{prefix}_pmc_summary.json", "w") as fh:
        json.dump(summary, fh, indent=1)
    if tj:
        print("summary written; merge the rows you need into", tj)


if __name__ == "__main__":
    main()
