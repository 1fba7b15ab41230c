# round 4: fresh processes, the state as the first allocation / behind a pad / at an offset of a larger allocation
set -o pipefail
O=gpurun_out/r4ax; mkdir -p $O; : > $O/pad.jsonl
for rep in 1 2 3 4; do
  for args in "first 0" "pad 16" "arena 16" "pad 32" "arena 32"; do
    timeout -k 10 120 python tools/pad_probe.py $args >> $O/pad.jsonl 2>> $O/pad.err || { echo "FAILED $args"; tail -3 $O/pad.err; exit 1; }
  done
done
cat $O/pad.jsonl
