# round 4: does the NUMA node of the page-locked state matter for a streamed run?  (half a config-5 rank slab, every row streamed)
set -o pipefail
O=gpurun_out/r4v; mkdir -p $O
export TVDN_STREAM_TIMING=1
echo "gpu numa: $(rocm-smi --showtoponuma 2>/dev/null | grep 'Numa Node')"
run() { name=$1; shift; timeout -k 10 300 python tools/stream_rates.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }; python - <<PY
import json
d=json.load(open("$O/$name.json"))
print("$name", {k:d.get(k) for k in ("value","value_whole_call","stream_rows","stream_k","resident_rows","passes","passes_s","setup_s","whole_call_s","h2d_GBps","d2h_GBps","skipped")})
PY
}
TVDN_NUMA=off run off_a 64x1024x256x256 2 38 76 0 &&
TVDN_NUMA=0 run node0 64x1024x256x256 2 38 76 0 &&
TVDN_NUMA=1 run node1 64x1024x256x256 2 38 76 0 &&
run auto 64x1024x256x256 2 38 76 0 &&
TVDN_NUMA=off run off_b 64x1024x256x256 2 38 76 0 &&
run auto_hybrid 64x1024x256x256 -1 -1 80 &&
TVDN_NUMA=off run off_hybrid 64x1024x256x256 -1 -1 80
