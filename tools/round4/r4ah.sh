# round 4: what a ring sweep costs by chunk height, no PCIe (every row resident): 32x1024x256x256, 8 levels per pass
set -o pipefail
O=gpurun_out/r4ah; mkdir -p $O
export TVDN_STREAM_TIMING=1
run() { name=$1; shift; timeout -k 10 300 python tools/stream_rates.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }; python - <<PY
import json
d=json.load(open("$O/$name.json"))
print("$name", {k:d.get(k) for k in ("value","stream_rows","stream_k","resident_rows","passes","passes_s","first_pass_s","value_later_passes")})
PY
}
run r1_k8 32x1024x256x256 1 8 32 32 &&
run r2_k8 32x1024x256x256 2 8 32 32 &&
run r4_k8 32x1024x256x256 4 8 32 32 &&
run r1_k16 32x1024x256x256 1 16 32 32 &&
run r2_k16 32x1024x256x256 2 16 32 32
