set -o pipefail
O=gpurun_out/r4n; mkdir -p $O
export TVDN_STREAM_TIMING=1
run() { name=$1; shift; timeout -k 10 200 python tools/stream_rates.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }; python - <<PY
import json
d=json.load(open("$O/$name.json"))
print("$name", {k:d.get(k) for k in ("value","value_whole_call","stream_rows","stream_k","resident_rows","passes","passes_s","setup_s","whole_call_s","h2d_GBps","d2h_GBps","skipped")})
PY
}
for db in 8 4 16; do
  TVDN_STREAM_DOWN_BLOCKS=$db run half_2_38_chain_db$db 64x1024x256x256 2 38 76 0
done
TVDN_STREAM_CHAIN=0 run half_2_38_drained 64x1024x256x256 2 38 76 0
TVDN_STREAM_CHAIN=0 TVDN_STREAM_DOWN_BLOCKS=8 run half_2_38_drained_db8 64x1024x256x256 2 38 76 0
run half_auto_chain 64x1024x256x256 -1 -1 80
TVDN_STREAM_CHAIN=0 run half_auto_drained 64x1024x256x256 -1 -1 80
run c2_16_64_chain 256x256x128x128 16 64 256 0
TVDN_STREAM_CHAIN=0 run c2_16_64_drained 256x256x128x128 16 64 256 0
run c2_16_128_chain 256x256x128x128 16 128 256 0
timeout -k 10 400 python -m pytest tests/test_gpu_run_streamed.py -x -q -p no:cacheprovider 2>&1 | tail -3
