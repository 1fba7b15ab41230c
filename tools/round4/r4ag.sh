# round 4: one-row chunks buy depth (rings of 3 rows instead of 4): PCIe-bound regime, every row streamed
set -o pipefail
O=gpurun_out/r4ag; mkdir -p $O
export TVDN_STREAM_TIMING=1
run() { name=$1; shift; timeout -k 10 300 python tools/stream_rates.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }; python - <<PY
import json
d=json.load(open("$O/$name.json"))
print("$name", {k:d.get(k) for k in ("value","stream_rows","stream_k","passes","passes_s","setup_s","whole_call_s","h2d_GBps","d2h_GBps","skipped")})
PY
}
run r1_k50 64x1024x256x256 1 50 150 0 &&
run r1_k48 64x1024x256x256 1 48 144 0 &&
run r2_k38 64x1024x256x256 2 38 152 0 &&
run r1_k44 64x1024x256x256 1 44 132 0
