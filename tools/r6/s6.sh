#!/bin/bash
# round 6, GPU session 6: chained launches of the streamed engine: bits (every streamed test) and the every-row-streamed rate with / without
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
echo "== tests"; timeout -k 10 1100 python -m pytest tests/test_gpu_ring.py tests/test_gpu_run_streamed.py tests/test_gpu_outofcore.py tests/test_gpu_two_ranks.py tests/test_gpu_rank_threads.py tests/test_gpu_nonfinite_wrap.py tests/test_gpu_pipelined.py -m gpu -x -q > $O/s6_tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -4 $O/s6_tests.log
[ $rc -eq 0 ] || exit 1
echo "== every row streamed, half a config-5 rank slab, three chained passes"
for v in 1 0 1 0; do
  TVDN_STREAM_CHAIN=$v TVDN_STREAM_TIMING=1 timeout -k 10 600 python tools/stream_rates.py 64x1024x256x256 -1 -1 -3 0 >> $O/s6_stream_chain_$v.jsonl 2>> $O/s6_stream_chain_$v.err; echo "chain=$v rc $?"; tail -1 $O/s6_stream_chain_$v.jsonl | cut -c1-500
done
