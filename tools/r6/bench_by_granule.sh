#!/bin/bash
# the bench headline (config 2, 20 steps, fresh process each) by granule size; then one slab of config 4
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for g in 1024 256 64; do
  TVDN_GRANULE_MIB=$g python3 $R/bench.py --no-api --no-also --no-cpu-baseline --no-sustained --audition-extra 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'granule_MiB': $g, 'rep': $rep, 'value': d['value'], 'kernel_ms': d['roofline']['kernel_ms'], 'min': d['roofline']['kernel_ms_min'], 'max': d['roofline']['kernel_ms_max'], 'state_mem': d['config']['state_mem']}))"
done; done
for g in 1024 256; do
  TVDN_GRANULE_MIB=$g python3 $R/bench.py --slab-of 8 --steps 10 --warmup 2 --no-api --no-also --no-cpu-baseline --no-sustained --audition-extra 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'slab_of_8': True, 'granule_MiB': $g, 'value': d['value'], 'kernel_ms': d['roofline']['kernel_ms'], 'state_mem': d['config']['state_mem']}))"
done
