#!/bin/bash
# small cubes by march length (TVDN_CHUNK): are they bound by the serial row steps of a march rather than by bytes?
R=${GRAFT_REPO_ROOT:-/root/repo}
S="32x32x128 64x64x256 96x96x384 128x128x512 256x256x256 16x16x64x64 32x32x64x64 64x64x64x64"
for c in default 1 2 4 8; do
  if [ $c = default ]; then unset TVDN_CHUNK; else export TVDN_CHUNK=$c; fi
  echo "== TVDN_CHUNK=$c"; python3 $R/tools/shape_sweep_probe.py $S 2>&1 | grep -v amdgpu
done
