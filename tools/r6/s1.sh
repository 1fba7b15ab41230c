#!/bin/bash
# round 6, GPU session 1: the hardened allocator's tests, the stress loops, a first-call breakdown, the 3-D plain A/B
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_vmm_guard.py tests/test_gpu_devmem.py tests/test_gpu_rccl.py tests/test_gpu_slabs.py -m gpu -x -q > $O/s1_tests.log 2>&1; echo "tests rc $?"; tail -5 $O/s1_tests.log
echo "== first call, fresh process x2 (TVDN_RUN_TIMING)"
for i in 1 2; do sleep 3; TVDN_RUN_TIMING=1 timeout -k 10 300 python tools/e2e_quick.py 256x256x128x128 50 3 > $O/s1_first_call_$i.jsonl 2> $O/s1_first_call_$i.err; echo "rc $?"; cat $O/s1_first_call_$i.jsonl | cut -c1-200; done
echo "== 3-D plain 512^3 A/B"
timeout -k 10 300 python tools/ab_inproc.py --config 3dplain --rounds 3 --steps 20 "base:" "chunk2:TVDN_CHUNK=2" "chunk4:TVDN_CHUNK=4" "chunk16:TVDN_CHUNK=16" "chunk32:TVDN_CHUNK=32" "xcd0:TVDN_XCD=0" "xcd0c4:TVDN_XCD=0;TVDN_CHUNK=4" "b128:TVDN_BLOCK=128" "b128c4:TVDN_BLOCK=128;TVDN_CHUNK=4" > $O/s1_plain3d_ab.jsonl 2> $O/s1_plain3d_ab.err; echo "rc $?"; cat $O/s1_plain3d_ab.jsonl | cut -c1-220
for shp in 1024x512x512 2048x512x512 512x1024x1024 2048x256x256 256x1024x1024; do
  timeout -k 10 300 python tools/ab_inproc.py --config 3dplain --shape $shp --rounds 2 --steps 10 "base:" "chunk4:TVDN_CHUNK=4" "chunk16:TVDN_CHUNK=16" "xcd0:TVDN_XCD=0" >> $O/s1_plain3d_shapes.jsonl 2>> $O/s1_plain3d_ab.err; echo "$shp rc $?"
done
cut -c1-200 $O/s1_plain3d_shapes.jsonl
timeout -k 10 300 python tools/ab_inproc.py --config 3d --rounds 2 --steps 20 "base:" "chunk4:TVDN_CHUNK=4" "chunk16:TVDN_CHUNK=16" "xcd0:TVDN_XCD=0" > $O/s1_fista3d_ab.jsonl 2>> $O/s1_plain3d_ab.err; cut -c1-200 $O/s1_fista3d_ab.jsonl
timeout -k 10 300 python tools/ab_inproc.py --config plain32 --rounds 2 --steps 10 "base:" "chunk4:TVDN_CHUNK=4" "chunk16:TVDN_CHUNK=16" > $O/s1_plain4d_ab.jsonl 2>> $O/s1_plain3d_ab.err; cut -c1-200 $O/s1_plain4d_ab.jsonl
echo "== stress"
timeout -k 10 900 python tools/vmm_stress.py --abort-cycles 50 --alloc-cycles 200 > $O/s1_vmm_stress.txt 2> $O/s1_vmm_stress.err; echo "stress rc $?"; grep -E "RESULT|CLEAN" $O/s1_vmm_stress.txt; tail -3 $O/s1_vmm_stress.err
