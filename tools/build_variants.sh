#!/bin/bash
# measurement helper: builds of libtvdn_hip.so with the streaming hints switched off, for A/B timing through
# TVDN_LIB=<path> (cytvdn_amd/_lib.py).  Output: tools/ubench/libtvdn_hip_<tag>.so (git-ignored).
set -e
cd "$(dirname "$0")/.."
SRC="cytvdn_amd/csrc/tvdn_capi.hip cytvdn_amd/csrc/tvdn_passes.hip cytvdn_amd/csrc/tvdn_fused.hip cytvdn_amd/csrc/tvdn_run.hip cytvdn_amd/csrc/tvdn_stream.hip cytvdn_amd/csrc/tvdn_hostio.hip"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -shared -Wl,-rpath,/opt/rocm/lib -lpthread"
build() { # tag, extra flags
  /opt/rocm/bin/hipcc $FLAGS $2 $SRC -o tools/ubench/libtvdn_hip_$1.so
}
build ntl0 "-DTVDN_NT_LOADS=0" &
build nts0 "-DTVDN_NT_STORES=0" &
build nt00 "-DTVDN_NT_LOADS=0 -DTVDN_NT_STORES=0" &
build mnt0 "-DTVDN_PASS_M_NT=0" &
build blk512 "-DTVDN_FUSED_BLOCK=512" &
build blk128 "-DTVDN_FUSED_BLOCK=128" &
wait
ls -la tools/ubench/*.so
