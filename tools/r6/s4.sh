#!/bin/bash
# round 6, GPU session 4: new allocator tests, first call with finer marks, the first-node kit rehearsed over gloo
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_devmem.py tests/test_gpu_vmm_guard.py tests/test_gpu_pipelined.py -m gpu -x -q > $O/s4_tests.log 2>&1; echo "tests rc $?"; tail -4 $O/s4_tests.log
for i in 1 2 3; do sleep 4; TVDN_RUN_TIMING=1 timeout -k 10 300 python tools/first_call_probe.py --reps 2 >> $O/s4_first_call.jsonl 2> $O/s4_first_call_$i.err; done
cut -c1-250 $O/s4_first_call.jsonl; grep -v "waited\|amdgpu.ids" $O/s4_first_call_1.err | head -24
sleep 4; TVDN_RUN_TIMING=1 timeout -k 10 300 python tools/first_call_probe.py --reps 3 --iters 200 >> $O/s4_first_call_200.jsonl 2> $O/s4_first_call_200.err; cut -c1-250 $O/s4_first_call_200.jsonl; grep "re-drawn\|granules of" $O/s4_first_call_200.err
echo "== kit rehearsal"
TVDN_DIST_BACKEND=gloo REHEARSE_RANKS="2" REHEARSE_SHAPE=1 timeout -k 10 900 bash tools/first_node_run.sh $O/first_node > $O/s4_kit.log 2>&1; echo "kit rc $?"; tail -25 $O/s4_kit.log
