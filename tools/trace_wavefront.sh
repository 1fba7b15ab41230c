#!/bin/bash
# Kernel + memory-copy timeline of one wavefront pass (host-resident config-2 cube); analysed by tools/trace_wavefront.py
# usage: tools/trace_wavefront.sh ROWS K [tag]
set -e
cd /tmp && export TMPDIR=/tmp
R=$1; K=$2; TAG=${3:-wf_${R}_${K}}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- \
    python3 $GRAFT_REPO_ROOT/tools/bench_outofcore.py --engine native --shape 256x256x128x128 --rows $R --k $K --iters $K \
    > $OUT/run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_wavefront.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
