#!/bin/bash
# Kernel + memory-copy timeline of denoise4D from NumPy (config 2, 50 iterations, three calls): how much of the uploads and
# downloads of tvdn_run's pipelined transfers runs under sweeps.  Condensed by tools/trace_wavefront.py.
#   gpurun -- 'bash tools/trace_pipelined.sh'          (TVDN_PIPELINE=0 bash tools/trace_pipelined.sh plain  for the plain order)
set -e
TAG=${1:-pipelined}
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- \
    python3 $GRAFT_REPO_ROOT/tools/e2e_quick.py 256x256x128x128 50 3 > $OUT/run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_wavefront.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
