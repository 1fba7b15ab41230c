# round 4: the planner with one-row chunks and the measured link model: its choices against round 4's earlier ones
set -o pipefail
O=gpurun_out/r4ai; mkdir -p $O
export TVDN_STREAM_TIMING=1
run() { name=$1; shift; timeout -k 10 300 python tools/stream_rates.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }; python - <<PY
import json
d=json.load(open("$O/$name.json"))
print("$name", {k:d.get(k) for k in ("value","value_later_passes","stream_rows","stream_k","resident_rows","passes","passes_s","setup_s","whole_call_s","h2d_GBps","d2h_GBps","skipped")})
PY
}
run hybrid_auto 64x1024x256x256 -1 -1 80 &&
run hybrid_old 64x1024x256x256 2 12 80 56 &&
run hybrid_r1_k12 64x1024x256x256 1 12 80 56 &&
run all_auto_3p 64x1024x256x256 -1 -1 -3 0 &&
run all_old_3p 64x1024x256x256 2 36 108 0
