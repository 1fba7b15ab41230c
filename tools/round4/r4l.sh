set -o pipefail
O=gpurun_out/r4l; mkdir -p $O
timeout -k 10 400 tools/ubench/pcie_duplex 12 engine 12 > $O/pcie_engine.jsonl 2>&1; cat $O/pcie_engine.jsonl
timeout -k 10 400 python -m pytest tests/test_gpu_run_streamed.py tests/test_gpu_pipelined.py tests/test_gpu_nonfinite_wrap.py tests/test_gpu_outofcore.py tests/test_gpu_cubeio.py -x -q -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
