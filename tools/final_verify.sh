#!/usr/bin/env bash
# End-of-round verification in ONE gpurun call: the -m gpu suite file by file (logs kept), the whole suite in one process as
# the driver runs it, __graft_entry__.smoke(), BASELINE config 1 on the box's host cores with the tracked CPU port.
#   gpurun --timeout 1100 -- 'bash tools/final_verify.sh'   then   python tools/final_verify_report.py > profiles/r03_gputest_head.txt
set -u -o pipefail
O=gpurun_out/r3final
rm -rf $O; mkdir -p $O
files="tests/test_gpu_kernel_level.py tests/test_gpu_hypothesis.py tests/test_gpu_parity.py tests/test_gpu_slabs.py tests/test_gpu_audition.py tests/test_gpu_pipelined.py tests/test_gpu_ring.py tests/test_gpu_run_streamed.py tests/test_gpu_nonfinite_wrap.py tests/test_gpu_cubeio.py tests/test_gpu_outofcore.py tests/test_gpu_bench_contract.py tests/test_gpu_two_ranks.py tests/test_gpu_rccl.py tests/test_gpu_fullsize.py tests/test_gpu_zz_misfit.py"
tools/gpu_suite_by_file.sh r3final 700 $files || exit 1
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/whole_suite.log 2>&1 || { tail -30 $O/whole_suite.log; exit 1; }
tail -2 $O/whole_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -3 $O/smoke.log
python tools/cpu_config1.py > $O/cpu_config1.json 2> $O/cpu_config1.err || { tail -5 $O/cpu_config1.err; exit 1; }
cat $O/cpu_config1.json
