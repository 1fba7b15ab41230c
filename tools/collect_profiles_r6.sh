#!/bin/bash
# round 6 evidence at HEAD (run through gpurun from the repo root): the default bench command as the driver runs it, the same
# command under rocprofv3 --kernel-trace --stats (split by grid: config 2 / config 3 / slab launches share one template), smoke(), BASELINE config 1 on the host cores with the tracked CPU port.  PMC traffic: collect_pmc_traffic.sh.
R=$(pwd)
O=$R/gpurun_out/r6prof
rm -rf $O; mkdir -p $O
set -o pipefail
git rev-parse HEAD > $O/commit.txt 2>/dev/null || cp .git_head $O/commit.txt 2>/dev/null || echo unknown > $O/commit.txt
python3 bench.py > $O/r06_bench_n1.json 2> $O/bench_n1.err || { echo "bench failed"; tail -5 $O/bench_n1.err; exit 1; }
echo "bench ok: $(head -c 200 $O/r06_bench_n1.json)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-api > $O/r06_a_bench_default_under_rocprof.json 2> $O/stats.log || { echo "rocprof stats failed"; tail -5 $O/stats.log; exit 1; }
echo "stats ok"
cd $R
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/r06_a_kernel_stats_bench_default.csv \;
find $O/stats -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace_full.csv \;
python3 - <<'PY'
import csv, collections, os
O = os.path.join(os.getcwd(), "gpurun_out", "r6prof")
acc = collections.defaultdict(list)
with open(os.path.join(O, "kernel_trace_full.csv"), newline="") as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"]
        if "tvdn::" in n:
            acc[(n.replace("void ", "").split("(")[0], r.get("Grid_Size") or r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open(os.path.join(O, "r06_a_kernel_trace_by_grid.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "grid", "calls", "mean_ms", "min_ms", "median_ms", "max_ms"])
    for (k, g), v in sorted(acc.items(), key=lambda t: -sum(t[1])):
        v = sorted(v)
        w.writerow([k, g, len(v), round(sum(v) / len(v), 4), round(v[0], 4), round(v[len(v) // 2], 4), round(v[-1], 4)])
PY
rm -rf $O/kernel_trace_full.csv $O/stats
head -8 $O/r06_a_kernel_trace_by_grid.csv
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -3 $O/smoke.log
python3 tools/cpu_config1.py > $O/cpu_config1.json 2> $O/cpu_config1.err || { tail -5 $O/cpu_config1.err; exit 1; }
cat $O/cpu_config1.json
