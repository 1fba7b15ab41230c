#!/bin/bash
# round 6, GPU session 5: does priming the link under the state's allocation pay?  (fresh processes, alternating)
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
for i in 1 2 3; do
  sleep 4; TVDN_RUN_TIMING=1 timeout -k 10 300 python tools/first_call_probe.py --reps 2 >> $O/s5_first_call_prime.jsonl 2> $O/s5_prime_$i.err
  sleep 4; TVDN_IO_PRIME=0 TVDN_RUN_TIMING=1 timeout -k 10 300 python tools/first_call_probe.py --reps 2 >> $O/s5_first_call_noprime.jsonl 2> $O/s5_noprime_$i.err
done
echo prime; cut -c1-200 $O/s5_first_call_prime.jsonl; echo noprime; cut -c1-200 $O/s5_first_call_noprime.jsonl
grep "rows .* up in\|upload +\|contexts\|state acq" $O/s5_prime_1.err | head -14
