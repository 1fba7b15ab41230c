#!/bin/bash
# round 2, third sweep: array skew against the plane shapes that were slow; pitch hypothesis (non-power-of-two B)
cd /root/repo
out=gpurun_out/tune_r2c.txt; : > $out
s() { lbl=$1; shift; envs=$1; shift
  r=$(env $envs python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])")
  echo "$lbl [$envs] $r" | tee -a $out; }
for sk in 0 4096 8192 16384 36864; do
  s slab8 TVDN_ARRAY_SKEW=$sk --slab-of 8
  s 128x256x256x128 TVDN_ARRAY_SKEW=$sk --shape 128x256x256x128
done
s 128x256x264x128 TVDN_ARRAY_SKEW=0 --shape 128x256x264x128
s 128x256x264x128 TVDN_ARRAY_SKEW=4096 --shape 128x256x264x128
s 64x512x256x264 TVDN_ARRAY_SKEW=0 --shape 64x512x256x264
s 64x512x256x264 TVDN_ARRAY_SKEW=4096 --shape 64x512x256x264
for sk in 0 2048 4096 8192 12288 20480; do s config2 TVDN_ARRAY_SKEW=$sk; done
s config3 TVDN_ARRAY_SKEW=0 --dtype f64 --plain
s config3 TVDN_ARRAY_SKEW=4096 --dtype f64 --plain
s config3 TVDN_ARRAY_SKEW=8192 --dtype f64 --plain
k() { lbl=$1; shift; echo "== $lbl" | tee -a $out; env "$@" python tools/time_kernel_level.py 2>/dev/null | grep f32 | tee -a $out; }
for ch in 8 16 32 64; do k "nts0 pass_chunk=$ch" TVDN_LIB=/root/repo/tools/ubench/libtvdn_hip_nts0.so TVDN_PASS_CHUNK=$ch; done
