#!/bin/bash
# the whole -m gpu suite once more (fresh box each gpurun call): how many clean suites since the pinned-lane fix
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
T=$(date +%H%M%S)
timeout -k 10 1150 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/suite_$T.log 2>&1; rc=$?
echo "suite rc $rc"; grep -v "socket.cpp\|Gloo\|amdgpu.ids" $O/suite_$T.log | tail -3
