#!/bin/bash
# round 6, GPU session 2: what the first call pays for (fresh processes, different warm-ups), plain-vs-granule 3-D plain, the rccl test again
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_two_ranks.py -m gpu -x -q > $O/s2_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/s2_tests.log
for w in "" lanes clock lanes,clock pool lanes,pool,clock; do
  sleep 3
  TVDN_RUN_TIMING=1 timeout -k 10 300 python tools/first_call_probe.py --warm "$w" --reps 2 >> $O/s2_first_call.jsonl 2> $O/s2_first_call_${w//,/_}.err; echo "warm=[$w] rc $?"
done
cut -c1-260 $O/s2_first_call.jsonl
echo "== 3-D plain 512^3 on plain hipMalloc blocks, 4 fresh processes (the r04 figure was 0.95 ms)"
for i in 1 2 3 4; do TVDN_VMM=0 timeout -k 10 200 python tools/ab_inproc.py --config 3dplain --rounds 2 --steps 20 "plain_block:" "chunk16:TVDN_CHUNK=16" >> $O/s2_plain3d_plainblocks.jsonl 2>/dev/null; done
for i in 1 2 3 4; do timeout -k 10 200 python tools/ab_inproc.py --config 3dplain --rounds 2 --steps 20 "granules:" "chunk16:TVDN_CHUNK=16" >> $O/s2_plain3d_granules.jsonl 2>/dev/null; done
cut -c1-200 $O/s2_plain3d_plainblocks.jsonl $O/s2_plain3d_granules.jsonl
