#!/bin/bash
# round 6, GPU session 12: the new pinned-copy test, then the round's evidence run (default bench, the same under rocprofv3, smoke, CPU config 1)
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_vmm_guard.py -m gpu -x -q > $O/s12_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/s12_tests.log
timeout -k 10 1000 bash tools/collect_profiles_r6.sh > $O/s12_collect.log 2>&1; echo "collect rc $?"; tail -15 $O/s12_collect.log
