#!/bin/bash
# round 3 evidence at HEAD (run through gpurun from the repo root): the default bench command as the driver runs it, the
# same command under rocprofv3 --kernel-trace --stats, separate PMC passes (FETCH_SIZE, WRITE_SIZE do not fit one pass;
# rocprofv3 gets the program itself after `--`), and a 600-step sustained run with the clocks sampled once per second.
R=$(pwd)
O=$R/gpurun_out/r3prof
mkdir -p $O
set -o pipefail
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { echo "bench failed"; tail -5 $O/bench_n1.err; exit 1; }
echo "bench ok: $(head -c 300 $O/bench_n1.json)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log || { echo "rocprof stats failed"; tail -5 $O/stats.log; exit 1; }
echo "stats ok"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-sustained > $O/pmc_fetch.json 2> $O/pmc_fetch.log || { echo "pmc fetch failed"; tail -5 $O/pmc_fetch.log; exit 1; }
echo "fetch ok"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-sustained > $O/pmc_write.json 2> $O/pmc_write.log || { echo "pmc write failed"; tail -5 $O/pmc_write.log; exit 1; }
echo "write ok"
cd $R
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/r03_b_bench > $O/r03_b_pmc_bench_summary.txt 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/r03_a_kernel_stats_bench_default.csv \;
find $O/stats -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace_full.csv \;
python3 - <<'PY'
import csv, collections, os
O = os.path.join(os.getcwd(), "gpurun_out", "r3prof")
acc = collections.defaultdict(list)
with open(os.path.join(O, "kernel_trace_full.csv"), newline="") as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"]
        if "tvdn::" in n:
            acc[(n.replace("void ", "").split("(")[0], r.get("Grid_Size") or r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open(os.path.join(O, "r03_a_kernel_trace_by_grid.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "grid", "calls", "mean_ms", "min_ms", "median_ms", "max_ms"])
    for (k, g), v in sorted(acc.items(), key=lambda t: -sum(t[1])):
        v = sorted(v)
        w.writerow([k, g, len(v), round(sum(v) / len(v), 4), round(v[0], 4), round(v[len(v) // 2], 4), round(v[-1], 4)])
PY
rm -f $O/kernel_trace_full.csv
python3 tools/clocks_during.py $O/r03_clocks_sustained.txt -- python3 bench.py --steps 600 --warmup 3 --no-also --no-sustained --no-cpu-baseline > $O/bench_600.json 2> $O/bench_600.err
echo "clocks rc=$?"
cat $O/r03_b_pmc_bench_summary.txt
head -8 $O/r03_a_kernel_trace_by_grid.csv
tail -2 $O/r03_clocks_sustained.txt | cut -c1-200
cat $O/bench_600.json | head -c 600
