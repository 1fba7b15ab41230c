#!/bin/bash
# round 6, GPU session 13: HBM-side counters per workload (separate --pmc passes), CPU config 1 with the port
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
python3 tools/cpu_config1.py > $O/s13_cpu_config1.json 2> $O/s13_cpu_config1.err; cat $O/s13_cpu_config1.json
timeout -k 10 1100 bash tools/collect_pmc_traffic.sh r06 > $O/s13_pmc.log 2>&1; echo "pmc rc $?"; tail -12 $O/s13_pmc.log
