#!/bin/bash
# round 6, GPU session 14: tests of what changed since the whole suite (bench watchdog, slab results through the lanes, fill beside the upload), API timing
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_pipelined.py tests/test_gpu_two_ranks.py tests/test_gpu_slabs.py tests/test_gpu_vmm_guard.py tests/test_gpu_devmem.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/s14_tests.log 2>&1; echo "tests rc $?"; tail -4 $O/s14_tests.log
for i in 1 2; do sleep 3; TVDN_RUN_TIMING=1 timeout -k 10 300 python tools/first_call_probe.py --reps 3 >> $O/s14_first_call.jsonl 2> $O/s14_first_call_$i.err; done
cut -c1-200 $O/s14_first_call.jsonl; grep -v "waited\|rows .* up\|amdgpu.ids" $O/s14_first_call_1.err | tail -9
