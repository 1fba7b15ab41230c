#!/bin/bash
# round 2, second sweep: one-pass kernels (store flavour x march length), fused sweep (array skew A/B, plane shape vs footprint)
cd /root/repo
out=gpurun_out/tune_r2b.txt; : > $out
k() { lbl=$1; shift; echo "== $lbl" | tee -a $out; env "$@" python tools/time_kernel_level.py 2>/dev/null | grep f32 | tee -a $out; }
for ch in 8 16 32 64; do k "nts0 pass_chunk=$ch" TVDN_LIB=/root/repo/tools/ubench/libtvdn_hip_nts0.so TVDN_PASS_CHUNK=$ch; done
k "nts1 pass_chunk=64" TVDN_PASS_CHUNK=64
b() { lbl=$1; shift
  r=$(env "$@" python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])")
  echo "fused $lbl $r" | tee -a $out; }
for rep in 1 2 3; do
  b "skew=0" X=1
  b "skew=4096" TVDN_ARRAY_SKEW=4096
  b "skew=1052672" TVDN_ARRAY_SKEW=1052672
done
s() { lbl=$1; shift
  r=$(python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])")
  echo "shape $lbl $r" | tee -a $out; }
s 256x256x128x128 --shape 256x256x128x128
s 32x512x256x256 --shape 32x512x256x256
s 64x512x256x256 --shape 64x512x256x256
s 512x256x128x128 --shape 512x256x128x128
s 128x256x256x128 --shape 128x256x256x128
s 128x512x128x128 --shape 128x512x128x128
