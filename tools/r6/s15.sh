#!/bin/bash
# round 6, GPU session 15: the first-node kit rehearsed over gloo at HEAD (2 and 4 ranks), the streamed handover A/B once more at HEAD
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
rm -rf $O/first_node
TVDN_DIST_BACKEND=gloo REHEARSE_RANKS="2 4" REHEARSE_SHAPE=1 timeout -k 10 900 bash tools/first_node_run.sh $O/first_node > $O/s15_kit.log 2>&1; echo "kit rc $?"; tail -14 $O/s15_kit.log
