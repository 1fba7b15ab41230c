#!/bin/bash
# round 6, GPU session 9: the whole -m gpu suite at this commit
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/s9_suite.log 2>&1; echo "suite rc $?"; tail -6 $O/s9_suite.log
