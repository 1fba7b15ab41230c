set -o pipefail
O=gpurun_out/r4y; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_run_streamed.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
export TVDN_STREAM_TIMING=1
run() { name=$1; shift; timeout -k 10 300 python tools/stream_rates.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }; grep "tvdn_run streamed" $O/$name.err; python - <<PY
import json
d=json.load(open("$O/$name.json"))
print("$name", {k:d.get(k) for k in ("value","value_whole_call","value_later_passes","first_pass_s","stream_k","resident_rows","passes","passes_s","setup_s","whole_call_s","h2d_GBps","d2h_GBps","skipped")})
PY
}
run hybrid_direct 64x1024x256x256 -1 -1 80 &&
TVDN_STREAM_HOME_AFTER=1 run hybrid_after 64x1024x256x256 -1 -1 80 &&
run hybrid_direct_b 64x1024x256x256 -1 -1 80
