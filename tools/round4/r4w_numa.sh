# round 4: page-locked state interleaved over the NUMA nodes vs first touch vs the device's node (every row streamed, 2 passes)
set -o pipefail
O=gpurun_out/r4w; mkdir -p $O
export TVDN_STREAM_TIMING=1
run() { name=$1; shift; timeout -k 10 300 python tools/stream_rates.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }; python - <<PY
import json
d=json.load(open("$O/$name.json"))
print("$name", {k:d.get(k) for k in ("value","value_whole_call","stream_k","resident_rows","passes","passes_s","setup_s","whole_call_s","h2d_GBps","d2h_GBps","skipped")})
PY
}
TVDN_NUMA=interleave run il_a 64x1024x256x256 2 38 76 0 &&
TVDN_NUMA=off run off_a 64x1024x256x256 2 38 76 0 &&
TVDN_NUMA=interleave run il_b 64x1024x256x256 2 38 76 0 &&
TVDN_NUMA=off run off_b 64x1024x256x256 2 38 76 0 &&
TVDN_NUMA=interleave run il_c 64x1024x256x256 2 38 76 0 &&
TVDN_NUMA=device run dev_a 64x1024x256x256 2 38 76 0
