#!/bin/bash
# measurement helper: sweep the fused-kernel tuning knobs on the GPU box (run through gpurun)
cd /root/repo
out=gpurun_out/tune_r1.txt; : > $out
run() { # label, env...
  lbl=$1; shift
  r=$(env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['check']['b_norm_last'])")
  echo "$lbl $r" | tee -a $out
}
run base X=1
for ch in 2 4 6 8 12; do run "chunk=$ch" TVDN_CHUNK=$ch; done
for ch in 4 8 32; do run "xcd=0 chunk=$ch" TVDN_XCD=0 TVDN_CHUNK=$ch; done
run "xcd=0 chunk=8 fake=7" TVDN_XCD=0 TVDN_CHUNK=8 TVDN_FAKE=7
run "chunk=4 fake=7" TVDN_CHUNK=4 TVDN_FAKE=7
