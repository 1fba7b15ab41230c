#!/bin/bash
# states of 30 and 60 GiB by granule size: fresh processes, kernel time of the FISTA f32 sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
S="256x128x128x128 256x256x128x128"
for g in default 64 128 256 512; do
  for rep in 1 2 3; do
    if [ $g = default ]; then unset TVDN_GRANULE_MIB; else export TVDN_GRANULE_MIB=$g; fi
    python3 $R/tools/shape_sweep_probe.py $S 2>&1 | grep -v amdgpu | sed "s/^{/{\"granule_MiB\": \"$g\", \"rep\": $rep, /"
  done
done
