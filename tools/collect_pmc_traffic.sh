#!/usr/bin/env bash
# HBM-side traffic of the fused sweep, workload by workload: ONE workload per process and ONE counter per pass (FETCH_SIZE and
# WRITE_SIZE do not fit one pass; rocprofv3 gets the program itself after `--`), so that every fused_iter_kernel row of a CSV
# belongs to the workload the JSON line of that process names.  tools/make_traffic_json.py turns the CSVs it leaves under
# profiles/ into profiles/traffic.json, byte for byte reproducibly.
#   bash tools/collect_pmc_traffic.sh r06      (on the GPU box; writes gpurun_out/pmc_<tag>/, then copy the condensed CSVs)
set -u -o pipefail
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 4 --warmup 1 --no-cpu-baseline --no-also --no-sustained --no-api --audition-extra 0"
run() {  # key, bench arguments...
  local key=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/${key}_$c" -- python3 "$R/bench.py" $COMMON "$@" \
      > "$O/${key}_$c.json" 2> "$O/${key}_$c.log" || { echo "$key $c failed"; tail -3 "$O/${key}_$c.log"; }
  done
  echo "$key done"
}
run config2
run config3 --dtype f64 --plain
run c1shape --shape 128x128x512
run slab8 --slab-of 8
run fista_f64 --dtype f64
run plain_f32 --plain
run plain3d_512 --shape 512x512x512 --plain
python3 "$R/tools/make_traffic_json.py" --collect "$O" --tag "$TAG" --out "$R/gpurun_out/pmc_$TAG/condensed"
