#!/bin/bash
# VERDICT r4 item 1, the "done" measurement: fresh processes on this box, each one (a) the bench headline on the product's state
# (granules) with the plain-hipMalloc placements of the same box beside it, (b) denoise4D NumPy -> NumPy at 50 iterations, first
# call of its process.  Appends one JSON object per process to gpurun_out/r5/placement_evidence.jsonl (tag = $1).
tag=${1:-box}
n=${2:-3}
out=gpurun_out/r5/placement_evidence_$tag.jsonl
mkdir -p gpurun_out/r5
for i in $(seq 1 $n); do
  sleep 5   # the memory the last process gave back is cleared in the background: let that finish
  timeout -k 10 300 python bench.py --no-also --no-api --no-cpu-baseline --no-sustained 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
h=d.get('hipmalloc_placements') or {}
print(json.dumps({'tag':'$tag','process':$i,'what':'bench headline','value':d['value'],'ms_per_step':d['ms_per_step'],'kernel_ms':d['roofline']['kernel_ms'],'state_mem':d['config']['state_mem'],
  'hipmalloc_best_of_4':h.get('value'),'hipmalloc_probe_ms':h.get('placement_audition_ms')}))" >> $out
  sleep 5
  timeout -k 10 300 python tools/e2e_quick.py 256x256x128x128 50 2 2>/dev/null | python -c "
import json,sys
r=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print(json.dumps({'tag':'$tag','process':$i,'what':'denoise4D NumPy -> NumPy, 50 iterations','first_call':r[0]['Gvoxel_iters_per_s_end_to_end'],'first_call_s':r[0]['seconds'],
  'second_call':r[1]['Gvoxel_iters_per_s_end_to_end'] if len(r)>1 else None}))" >> $out
  tail -2 $out
done
