#!/bin/bash
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_vmm_guard.py -m gpu -x -q > $O/s16_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/s16_tests.log
for w in api "" api ""; do sleep 3; TVDN_RUN_TIMING=1 timeout -k 10 300 python tools/first_call_probe.py --warm "$w" --reps 2 >> $O/s16_first_call.jsonl 2> $O/s16_first_call_err.txt; done
cut -c1-230 $O/s16_first_call.jsonl
