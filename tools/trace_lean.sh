#!/bin/bash
# rocprofv3 --kernel-trace --stats of the streamed engine keeping every row of half a config-5 rank slab in HBM (lean layout, rows
# swept in place): the ring instantiation's launches by grid, to set beside the resident sweep's (profiles/r05_lean_kernel_trace_by_grid.csv)
R=$(pwd)
O=$R/gpurun_out/r5lean
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/ubench/resident_rows_probe.py shapes 8:5 > $O/run.log 2> $O/stats.log || { tail -5 $O/stats.log; exit 1; }
cd $R
find $O/stats -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace_full.csv \;
python3 - <<'PY'
import csv, collections, os
O = os.path.join(os.getcwd(), "gpurun_out", "r5lean")
acc = collections.defaultdict(list)
with open(os.path.join(O, "kernel_trace_full.csv"), newline="") as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"]
        if "tvdn::" in n:
            acc[(n.replace("void ", "").split("(")[0], r.get("Grid_Size") or r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open(os.path.join(O, "r05_lean_kernel_trace_by_grid.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "grid", "calls", "mean_ms", "min_ms", "median_ms", "max_ms", "total_ms"])
    for (k, g), v in sorted(acc.items(), key=lambda t: -sum(t[1])):
        v = sorted(v)
        w.writerow([k, g, len(v), round(sum(v) / len(v), 4), round(v[0], 4), round(v[len(v) // 2], 4), round(v[-1], 4), round(sum(v), 1)])
PY
rm -rf $O/kernel_trace_full.csv $O/stats
head -8 $O/r05_lean_kernel_trace_by_grid.csv; grep "^{" $O/run.log | cut -c1-260
