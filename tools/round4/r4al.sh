set -o pipefail
O=gpurun_out/r4al; mkdir -p $O
TVDN_PIN_IN_PLACE_MIN=64K timeout -k 10 600 python -m pytest tests/test_gpu_pipelined.py tests/test_gpu_hypothesis.py -x -q > $O/tests_lowmin.log 2>&1 || { tail -30 $O/tests_lowmin.log; exit 1; }
tail -2 $O/tests_lowmin.log
timeout -k 10 600 python -m pytest tests/test_gpu_pipelined.py tests/test_gpu_fullsize.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
TVDN_RUN_TIMING=1 timeout -k 10 300 python tools/e2e_quick.py 256x256x128x128 50 3 > $O/e2e_50.txt 2>&1 || { tail $O/e2e_50.txt; exit 1; }
grep "copied in\|download\|Gvoxel" $O/e2e_50.txt | tail -13
TVDN_RESULT_LANES=1 timeout -k 10 300 python tools/e2e_quick.py 256x256x128x128 50 3 > $O/e2e_50_lanes.txt 2>&1
grep "Gvoxel" $O/e2e_50_lanes.txt
timeout -k 10 300 python tools/e2e_quick.py 256x256x128x128 200 2 > $O/e2e_200.txt 2>&1
grep "Gvoxel" $O/e2e_200.txt
TVDN_RESULT_LANES=1 timeout -k 10 300 python tools/e2e_quick.py 256x256x128x128 200 2 > $O/e2e_200_lanes.txt 2>&1
grep "Gvoxel" $O/e2e_200_lanes.txt
