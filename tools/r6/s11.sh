#!/bin/bash
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
AMD_LOG_LEVEL=4 timeout -k 10 300 python tools/ubench/pin_cache_probe.py 2> /tmp/pin_probe.log; echo "probe rc $?"; wc -l /tmp/pin_probe.log
python3 tools/ubench/pin_cache_probe.py --digest /tmp/pin_probe.log > $O/s11_pin_cache_probe.jsonl; cat $O/s11_pin_cache_probe.jsonl
grep -i "pinned\|staged" /tmp/pin_probe.log | sed 's/^.*\] //' | cut -c1-160 | sort | uniq -c | sort -rn | head -12
