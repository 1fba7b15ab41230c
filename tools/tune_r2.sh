#!/bin/bash
# measurement helper (run through gpurun): one-pass kernel knobs and fused-sweep knobs, round 2
cd /root/repo
out=gpurun_out/tune_r2.txt; : > $out
k() { # label, env...
  lbl=$1; shift
  echo "== $lbl" | tee -a $out
  env "$@" python tools/time_kernel_level.py 2>/dev/null | grep f32 | tee -a $out
}
k base X=1
for ch in 2 4 16 32; do k "pass_chunk=$ch" TVDN_PASS_CHUNK=$ch; done
for v in ntl0 nts0 nt00; do k "variant=$v" TVDN_LIB=/root/repo/tools/ubench/libtvdn_hip_$v.so; done
b() { # label, env...
  lbl=$1; shift
  r=$(env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['roofline']['kernel_ms'], d['check']['b_norm_last'])")
  echo "fused $lbl $r" | tee -a $out
}
b base X=1
for sk in 256 4096 69632 1052672; do b "skew=$sk" TVDN_ARRAY_SKEW=$sk; done
b base2 X=1
s() { # label, env...
  lbl=$1; shift
  r=$(env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --slab-of 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['roofline']['kernel_ms'], d['ms_per_step'])")
  echo "slab $lbl $r" | tee -a $out
}
for e in 1 4 8 16; do s "edge_rows=$e" TVDN_EDGE_ROWS=$e; done
