#!/bin/bash
# round 6, GPU session 3: parallel hipMemCreate?  first call with the new lanes, by spread budget; arrangement search; stress again (another box)
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
echo "== parallel hipMemCreate (fresh device)"; timeout -k 10 200 tools/ubench/vmm_create_parallel 96 1024 > $O/s3_create_parallel.jsonl 2>&1; cat $O/s3_create_parallel.jsonl
sleep 5
echo "== first call by spread budget (fresh processes)"
for b in default 0.1 0.03 0; do
  for i in 1 2; do
    sleep 4
    if [ "$b" = default ]; then TVDN_RUN_TIMING=1 timeout -k 10 300 python tools/first_call_probe.py --reps 2 >> $O/s3_first_call_$b.jsonl 2>> $O/s3_first_call_$b.err
    else TVDN_SPREAD_S=$b TVDN_RUN_TIMING=1 timeout -k 10 300 python tools/first_call_probe.py --reps 2 >> $O/s3_first_call_$b.jsonl 2>> $O/s3_first_call_$b.err; fi
  done
  echo "budget $b"; cut -c1-250 $O/s3_first_call_$b.jsonl
done
echo "== arrangement search"
for i in 1 2 3; do sleep 4; timeout -k 10 300 python tools/arrangement_search.py --deals 8 >> $O/s3_arrangement_search.jsonl 2>/dev/null; done
for i in 1 2; do sleep 4; timeout -k 10 300 python tools/arrangement_search.py --deals 6 --fresh >> $O/s3_arrangement_search.jsonl 2>/dev/null; done
cut -c1-400 $O/s3_arrangement_search.jsonl
echo "== stress"
timeout -k 10 900 python tools/vmm_stress.py --abort-cycles 100 --alloc-cycles 300 > $O/s3_vmm_stress.txt 2> $O/s3_vmm_stress.err; echo "stress rc $?"; grep -E "RESULT|CLEAN|^#" $O/s3_vmm_stress.txt; tail -3 $O/s3_vmm_stress.err
