#!/bin/bash
# the sweep on mid-size 4-D cubes by march length (TVDN_CHUNK) and state memory
R=${GRAFT_REPO_ROOT:-/root/repo}
S="32x32x128x128 64x64x128x128 128x64x128x128 128x128x128x128 256x128x128x128 256x256x128x128"
for c in 8 4 2 16; do echo "== TVDN_CHUNK=$c"; TVDN_CHUNK=$c python3 $R/tools/shape_sweep_probe.py $S 2>&1 | grep -v amdgpu; done
echo "== TVDN_VMM=0 (plain blocks)"; TVDN_VMM=0 python3 $R/tools/shape_sweep_probe.py $S 2>&1 | grep -v amdgpu
echo "== TVDN_XCD=0"; TVDN_XCD=0 python3 $R/tools/shape_sweep_probe.py $S 2>&1 | grep -v amdgpu
