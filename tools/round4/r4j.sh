set -o pipefail
O=gpurun_out/r4j; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_run_streamed.py tests/test_gpu_pipelined.py tests/test_gpu_nonfinite_wrap.py tests/test_gpu_outofcore.py tests/test_gpu_cubeio.py -x -q -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/tests.log
export TVDN_STREAM_TIMING=1
run() { name=$1; shift; timeout -k 10 200 python tools/stream_rates.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }; grep "tvdn_run streamed: rows" $O/$name.err | cut -c1-200; python - <<PY
import json
d=json.load(open("$O/$name.json"))
print("$name", {k:d.get(k) for k in ("value","value_whole_call","stream_rows","stream_k","resident_rows","passes","passes_s","setup_s","whole_call_s","h2d_GBps","d2h_GBps","skipped")})
PY
}
run half_2_38_chain 64x1024x256x256 2 38 76 0 &&
TVDN_STREAM_CHAIN=0 run half_2_38_drained 64x1024x256x256 2 38 76 0 &&
run half_auto_chain 64x1024x256x256 -1 -1 80 &&
TVDN_STREAM_CHAIN=0 run half_auto_drained 64x1024x256x256 -1 -1 80 &&
run c2_16_64_chain 256x256x128x128 16 64 256 0 &&
TVDN_STREAM_CHAIN=0 run c2_16_64_drained 256x256x128x128 16 64 256 0
timeout -k 10 200 python tools/ceiling_vs_sweep.py --config 2 --hold 2 --drop-recon > $O/ceiling_drop_recon.jsonl 2> $O/ceiling.err; cat $O/ceiling_drop_recon.jsonl | cut -c1-400
