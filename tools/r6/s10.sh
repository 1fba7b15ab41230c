#!/bin/bash
# round 6, GPU session 10: the pin-cache probe, then the whole -m gpu suite
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
timeout -k 10 200 python tools/ubench/pin_cache_probe.py > $O/s10_pin_cache_probe.jsonl 2> $O/s10_probe.err; echo "probe rc $?"; cat $O/s10_pin_cache_probe.jsonl
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=12 > $O/s10_suite.log 2>&1; echo "suite rc $?"; tail -22 $O/s10_suite.log
