set -o pipefail
O=gpurun_out/r4o; mkdir -p $O
export TVDN_STREAM_TIMING=1
run() { name=$1; shift; timeout -k 10 300 python tools/stream_rates.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }; python - <<PY
import json
d=json.load(open("$O/$name.json"))
print("$name", {k:d.get(k) for k in ("value","value_whole_call","stream_rows","stream_k","resident_rows","passes","passes_s","setup_s","whole_call_s","h2d_GBps","d2h_GBps","skipped")})
PY
}
run half_2_38x4_chain 64x1024x256x256 2 38 152 0
TVDN_STREAM_CHAIN=0 run half_2_38x4_drained 64x1024x256x256 2 38 152 0
TVDN_STREAM_DOWN_BLOCKS=0 run half_2_38x4_chain_memcpy 64x1024x256x256 2 38 152 0
