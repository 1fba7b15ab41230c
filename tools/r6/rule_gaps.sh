#!/bin/bash
# the GPU's idle time between consecutive sweeps of a run with a stopping rule: sums read in stream order (TVDN_STOP_LAG=0) against one
# iteration behind (default), from rocprofv3 --kernel-trace (the profiler inflates both; the difference is what counts)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for lag in 0 1; do for shp in 64x64x256 128x128x512; do
  rm -rf /tmp/pg
  if [ $lag = 0 ]; then export TVDN_STOP_LAG=0; else unset TVDN_STOP_LAG; fi
  rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -- python3 $R/tools/r6/small_loop.py $shp 1000 rule > /tmp/pg.out 2>/dev/null
  python3 - <<PY
import csv, glob, json
f = glob.glob("/tmp/pg/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
sw = [(s, e) for s, e, n in rows if "fused_iter_kernel" in n][30:]           # (past the warm-up call)
gaps = [(sw[i + 1][0] - sw[i][1]) / 1e3 for i in range(len(sw) - 1)]
gaps.sort()
dur = sorted((e - s) / 1e3 for s, e in sw)
print(json.dumps({"shape": "$shp", "TVDN_STOP_LAG": $lag, "sweeps": len(sw), "sweep_us_median": round(dur[len(dur) // 2], 2),
                  "gap_between_sweeps_us_median": round(gaps[len(gaps) // 2], 2), "gap_us_p10": round(gaps[len(gaps) // 10], 2), "gap_us_p90": round(gaps[len(gaps) * 9 // 10], 2),
                  "wall": open("/tmp/pg.out").read().strip()[-40:]}))
PY
done; done
