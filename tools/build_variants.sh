#!/bin/bash
# measurement helper: builds of libtvdn_hip.so with the streaming hints switched off, for A/B timing through
# TVDN_LIB=<path> (cytvdn_amd/_lib.py).  Output: tools/ubench/libtvdn_hip_<tag>.so (git-ignored).
set -e
cd "$(dirname "$0")/.."
SRC="$(ls cytvdn_amd/csrc/*.hip | tr "\n" " ")"   # every source of the library (cytvdn_amd/csrc/Makefile)
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -shared -Wl,-rpath,/opt/rocm/lib -lpthread"
build() { # tag, extra flags
  /opt/rocm/bin/hipcc $FLAGS $2 $SRC -o tools/ubench/libtvdn_hip_$1.so
}
if [ "$1" = "blocks" ]; then   # workgroup sizes of the fused sweep, for tools/ab_inproc.py
  for b in 128 512 1024; do build blk$b "-DTVDN_FUSED_BLOCK=$b" & done
  build ntm4 "-DTVDN_NTMASK=4" &
  wait; ls -la tools/ubench/*blk*.so; exit 0
fi
if [ "$1" = "waves" ]; then   # round 4: occupancy targets of the fused sweep (amdgpu_waves_per_eu)
  for w in 3 4 5 6 7 8; do build wav$w "-DTVDN_WAVES_PER_EU=$w" & done
  wait; ls -la tools/ubench/*wav*.so; exit 0
fi
if [ "$1" = "aux" ]; then   # round 6: cache-policy bits of the sweep's buffer accesses (sc0 = 1, nt = 2, sc1 = 16)
  for v in 18 19 16 3; do build st$v "-DTVDN_ST_AUX=$v -DTVDN_ST_AUX64=$v" & done
  wait
  for v in 18 19 3; do build ldnt$v "-DTVDN_LDNT_AUX=$v" & done
  build ld1 "-DTVDN_LD_AUX=1" &
  wait; ls -la tools/ubench/*st1*.so tools/ubench/*ld*.so; exit 0
fi
if [ "$1" = "order" ]; then   # round 6: the data term's load first instead of last
  build origfirst "-DTVDN_ORIG_FIRST=1"; ls -la tools/ubench/*origfirst*.so; exit 0
fi
if [ "$1" = "ntmask" ]; then   # round 3: which accumulator-state loads stream past the L2 (csrc/tvdn_fused.hip, kNtMask)
  for m in 1 2 3 7 11 27 31 4 16; do build ntm$m "-DTVDN_NTMASK=$m" & done
  wait; ls -la tools/ubench/*ntm*.so; exit 0
fi
build ntl0 "-DTVDN_NT_LOADS=0" &
build nts0 "-DTVDN_NT_STORES=0" &
build nt00 "-DTVDN_NT_LOADS=0 -DTVDN_NT_STORES=0" &
build mnt0 "-DTVDN_PASS_M_NT=0" &
build blk512 "-DTVDN_FUSED_BLOCK=512" &
build blk128 "-DTVDN_FUSED_BLOCK=128" &
wait
ls -la tools/ubench/*.so
