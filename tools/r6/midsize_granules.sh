#!/bin/bash
# mid-size states (2 ... 16 GiB) by granule size: fresh processes (each draws its own placement), kernel time of the FISTA f32 sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
S="48x48x128x128 64x64x128x128 128x64x128x128 128x128x128x128"
for g in default 64 128 256 512; do
  for rep in 1 2 3 4; do
    if [ $g = default ]; then unset TVDN_GRANULE_MIB; else export TVDN_GRANULE_MIB=$g; fi
    python3 $R/tools/shape_sweep_probe.py $S 2>&1 | grep -v amdgpu | sed "s/^{/{\"granule_MiB\": \"$g\", \"rep\": $rep, /"
  done
done
export TVDN_VMM=0; unset TVDN_GRANULE_MIB
for rep in 1 2 3 4; do python3 $R/tools/shape_sweep_probe.py $S 2>&1 | grep -v amdgpu | sed "s/^{/{\"granule_MiB\": \"plain\", \"rep\": $rep, /"; done
