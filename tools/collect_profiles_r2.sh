#!/bin/bash
# round 2 evidence (run through gpurun): rocprofv3 kernel stats of the exact default bench command, and separate PMC
# passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) for the bench workloads and for the one-pass kernels.
R=/root/repo
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_r2 $O/pmc_r2_fetch $O/pmc_r2_write $O/pmc_r2_kl_fetch $O/pmc_r2_kl_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r2 -- python3 $R/bench.py > $O/prof_r2_bench.json 2> $O/prof_r2.log
echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_r2_fetch -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_r2_fetch.json 2> $O/pmc_r2_fetch.log
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_r2_write -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_r2_write.json 2> $O/pmc_r2_write.log
echo "write rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_r2_kl_fetch -- python3 $R/tools/time_kernel_level.py > $O/pmc_r2_kl_fetch.jsonl 2> $O/pmc_r2_kl_fetch.log
echo "kl fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_r2_kl_write -- python3 $R/tools/time_kernel_level.py > $O/pmc_r2_kl_write.jsonl 2> $O/pmc_r2_kl_write.log
echo "kl write rc=$?"
cd $R
python3 tools/pmc_summary.py $O/pmc_r2_fetch $O/pmc_r2_write $O/r2_bench > $O/r2_bench_pmc.txt 2>&1
python3 tools/pmc_summary.py $O/pmc_r2_kl_fetch $O/pmc_r2_kl_write $O/r2_kl > $O/r2_kl_pmc.txt 2>&1
find $O/prof_r2 -name "*kernel_stats.csv" -exec cp {} $O/r2_kernel_stats_bench_default.csv \;
cat $O/r2_bench_pmc.txt $O/r2_kl_pmc.txt
head -12 $O/r2_kernel_stats_bench_default.csv
