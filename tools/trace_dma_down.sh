#!/bin/bash
# Where do the runtime's downloads lose their time inside chained passes?  Every row of half a config-5 rank slab streamed, three
# chained passes of 49 levels, downloads as DOWN_BLOCKS says (unset: the default, the DMA pump), under rocprofv3 --memory-copy-trace
# --kernel-trace: duration of every copy by direction and size, and which kernels ran (blit kernels of the runtime included).
R=$(pwd)
O=$R/gpurun_out/r5dma
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PROBE_SKIP_PLAN=1 PROBE_RESIDENT=0 PROBE_ITERS=147
[ -n "${DOWN_BLOCKS:-}" ] && export TVDN_STREAM_DOWN_BLOCKS=$DOWN_BLOCKS   # 0: the runtime's copies queued behind events; 8: the copy kernel; unset: the DMA pump (default)
timeout -k 10 400 rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/ubench/resident_rows_probe.py shapes 1:49 > $O/run.log 2> $O/trace.log || { tail -5 $O/trace.log; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob, os, collections, json
O = os.path.join(os.getcwd(), "gpurun_out", "r5dma")
cp = glob.glob(os.path.join(O, "trace", "**", "*memory_copy_trace.csv"), recursive=True)
kt = glob.glob(os.path.join(O, "trace", "**", "*kernel_trace.csv"), recursive=True)
out = {}
if cp:
    rows = list(csv.DictReader(open(cp[0], newline="")))
    print("copy columns:", list(rows[0].keys()))
    by = collections.defaultdict(list)
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        by[r.get("Direction") or r.get("Kind") or "?"].append(d)
    for k, v in by.items():
        v.sort()
        print(k, len(v), "median %.2f ms  p90 %.2f  max %.2f  total %.1f ms" % (v[len(v) // 2], v[int(len(v) * 0.9)], v[-1], sum(v)))
if kt:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0], newline="")):
        acc[r["Kernel_Name"].split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k, v in sorted(acc.items(), key=lambda t: -sum(t[1]))[:8]:
        print("%-72s %6d calls  mean %.3f ms  total %.1f ms" % (k, len(v), sum(v) / len(v), sum(v)))
PY
grep "^{" $O/run.log | tail -1 | cut -c1-300
rm -rf $O/trace
