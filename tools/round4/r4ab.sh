# round 4: workgroups of the download copy kernel in the PCIe-bound regime (every row streamed, four chained passes)
set -o pipefail
O=gpurun_out/r4ab; mkdir -p $O
export TVDN_STREAM_TIMING=1
run() { name=$1; shift; timeout -k 10 300 python tools/stream_rates.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }; python - <<PY
import json
d=json.load(open("$O/$name.json"))
print("$name", {k:d.get(k) for k in ("value","stream_k","passes","passes_s","setup_s","whole_call_s","h2d_GBps","d2h_GBps","skipped")})
PY
}
TVDN_STREAM_DOWN_BLOCKS=8 run db8 64x1024x256x256 2 38 152 0 &&
TVDN_STREAM_DOWN_BLOCKS=32 run db32 64x1024x256x256 2 38 152 0 &&
TVDN_STREAM_DOWN_BLOCKS=64 run db64 64x1024x256x256 2 38 152 0 &&
TVDN_STREAM_DOWN_BLOCKS=16 run db16 64x1024x256x256 2 38 152 0 &&
TVDN_STREAM_DOWN_BLOCKS=128 run db128 64x1024x256x256 2 38 152 0
