set -o pipefail
O=gpurun_out/r4d; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_run_streamed.py tests/test_gpu_pipelined.py tests/test_gpu_nonfinite_wrap.py tests/test_gpu_outofcore.py tests/test_gpu_cubeio.py -x -q -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
export TVDN_STREAM_TIMING=1
run() { name=$1; shift; timeout -k 10 200 python tools/stream_rates.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }; grep "tvdn_run streamed" $O/$name.err; python - <<PY
import json
d=json.load(open("$O/$name.json"))
print("$name", {k:d.get(k) for k in ("value","value_whole_call","stream_rows","stream_k","resident_rows","passes","passes_s","setup_s","whole_call_s","h2d_GBps","d2h_GBps","skipped")})
PY
}
run half_auto 64x1024x256x256 -1 -1 80 &&
run half_2_40_res0 64x1024x256x256 2 40 80 0 &&
run q_2_8 32x1024x256x256 2 8 16 32 &&
run q_4_8 32x1024x256x256 4 8 16 32 &&
run q_8_4 32x1024x256x256 8 4 16 32 &&
run q_2_4 32x1024x256x256 2 4 16 32 &&
run q_4_4 32x1024x256x256 4 4 16 32 &&
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_q_2_8 -- python3 $GRAFT_REPO_ROOT/tools/stream_rates.py 32x1024x256x256 2 8 16 32 > $GRAFT_REPO_ROOT/$O/prof_q_2_8.json 2> $GRAFT_REPO_ROOT/$O/prof_q_2_8.err; echo "prof rc=$?")
find $O/prof_q_2_8 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {}'
