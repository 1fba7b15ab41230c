#!/bin/bash
# round 6, GPU session 7: the accumulator handed across the cut between launches (TVDN_STREAM_HANDOVER), every row streamed, A/B
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
for v in 1 0 1 0; do
  TVDN_STREAM_HANDOVER=$v TVDN_STREAM_TIMING=1 timeout -k 10 600 python tools/stream_rates.py 64x1024x256x256 -1 -1 -3 0 > $O/s7_one.json 2>> $O/s7_handover_$v.err; echo "handover=$v rc $?"
  python3 -c "
import json; d=json.load(open('$O/s7_one.json')); d['TVDN_STREAM_HANDOVER']=$v; print(json.dumps(d))" >> $O/s7_handover.jsonl; tail -1 $O/s7_handover.jsonl | cut -c1-420
done
# two-row chunks for comparison (the planner's other candidate)
for v in 1 0; do
  TVDN_STREAM_HANDOVER=$v timeout -k 10 600 python tools/stream_rates.py 64x1024x256x256 2 37 -3 0 > $O/s7_one.json 2>> $O/s7_handover_$v.err; echo "R2 handover=$v rc $?"
  python3 -c "
import json; d=json.load(open('$O/s7_one.json')); d['TVDN_STREAM_HANDOVER']=$v; print(json.dumps(d))" >> $O/s7_handover.jsonl; tail -1 $O/s7_handover.jsonl | cut -c1-420
done
