#!/usr/bin/env bash
# One-shot kit for the FIRST run on a multi-GPU MI355X node (no round so far had one: RCCL has never executed, SURVEY 8e /
# BASELINE configs[3] and [4] have no hardware number).  Every step runs in its own process under its own timeout with its log
# kept, a failing step does not stop the later ones, and nothing here starts more ranks than the node has GPUs:
#   1. tests/test_gpu_rccl.py            the RCCL data path against the oracle (2 and 3 ranks, chain / ring, overlapped / blocking)
#   2. bench.py --gpus 2, 4, 8           weak scaling of BASELINE configs[3] (one 64-row slab of 512x256x256 planes per GPU),
#                                        each with its pre-flight (bit-exact exchange, ranks seen == world, one GPU per rank)
#   3. tools/bench_staged_slabs.py       BASELINE configs[4] in structure: every rank streams its slab from page-locked host
#                                        memory (needs 40 GiB of host memory per rank at the default shape)
#   4. tools/device_list_streamed.py     the same structure inside ONE process: tvdn_run with the node's GPUs as a device list,
#                                        every slab streamed from host arrays the slabs share (no launcher, no messages)
#   5. tools/device_list_resident.py     resident slabs inside ONE process (tvdn_run with a device list, halo rows by peer copies),
#                                        twice: slabs on granules with every device of the list granted access (ABI 9; the run
#                                        checks a peer copy out of every block first: peer_check) and TVDN_VMM_PEER=0 (plain
#                                        hipMalloc blocks) -- the A/B of placement under peer copies
#   6. tools/first_node_report.py        one page: what ran, what it gave, scaling efficiency against the 1-GPU line
# Since round 5 a state of 2 GiB or more lives on 1 GiB granules of HIP virtual memory (csrc/tvdn_devmem.hip) -- memory RCCL has
# never been handed on hardware either.  The 2-GPU bench therefore runs twice, the second time with TVDN_VMM=0 (plain hipMalloc):
# if only the first fails, the granules are the cause and TVDN_VMM=0 is the way round it; if both run, the pair is the first
# measurement of placement under an exchange.  (bench.py also repeats its pre-flight with the small states on granules and falls
# back to plain memory by itself when that one gives wrong bits or an error: preflight.on_granules / state_mem_fallback in its line.)
# Usage (on the node, from the repo root):   bash tools/first_node_run.sh [OUTDIR]
# Rehearsal on a one-GPU box (several ranks share the GPU, halo rows staged through host memory over gloo; at most 6 processes
# may use a GPU at once on the pool's boxes, launcher included: 4 ranks is the most that rehearses safely):   TVDN_DIST_BACKEND=gloo REHEARSE_RANKS="2 4" REHEARSE_SHAPE=1 bash tools/first_node_run.sh
set -u -o pipefail
O=${1:-gpurun_out/first_node}
mkdir -p "$O"
: > "$O/summary.txt"
NGPU=$(python3 -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
echo "GPUs visible: $NGPU" | tee -a "$O/summary.txt"
export HSA_ENABLE_IPC_MODE_LEGACY=0
step() {  # name, timeout seconds, command...
  local name=$1 t=$2; shift 2
  local t0=$SECONDS
  timeout -k 10 "$t" "$@" > "$O/$name.out" 2> "$O/$name.err"
  local rc=$?
  echo "$name rc=$rc $((SECONDS - t0)) s" | tee -a "$O/summary.txt"
}
RANKS=${REHEARSE_RANKS:-"2 4 8"}
step rccl_tests 900 python3 -m pytest tests/test_gpu_rccl.py -q -p no:cacheprovider
step bench_gpus_1 900 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-also --no-sustained --no-api --audition-extra 0 --no-cpu-baseline
for n in $RANKS; do
  if [ -n "${REHEARSE_SHAPE:-}" ]; then
    # rehearsal: a slab of 2 rows of small planes per rank, the ranks share the GPU(s) that exist
    step bench_gpus_$n 300 python3 bench.py --gpus "$n" --steps 6 --warmup 2 --shape $((2 * n))x64x64x64
  elif [ "$n" -le "$NGPU" ]; then
    step bench_gpus_$n 900 python3 bench.py --gpus "$n" --steps 10 --warmup 3
    if [ "$n" -eq 2 ]; then TVDN_VMM=0 step bench_gpus_2_plain 900 python3 bench.py --gpus 2 --steps 10 --warmup 3; fi
  else
    echo "bench_gpus_$n skipped: $NGPU GPUs" | tee -a "$O/summary.txt"
  fi
done
if [ -n "${REHEARSE_SHAPE:-}" ]; then
  step staged_slabs 300 python3 tools/bench_staged_slabs.py --shape 32x64x64x64 --ranks 2 --rows 4 --k 4 --iters 8
  step device_list_streamed 300 python3 tools/device_list_streamed.py --shape 32x64x64x64 --devices 0,0,0 --rows 4 --k 4 --iters 8 --check
  TVDN_PEER_CHECK=1 TVDN_VMM_MIN_MIB=1 TVDN_GRANULE_MIB=2 step device_list_resident 300 python3 tools/device_list_resident.py --shape 32x64x64x64 --devices 0,0,0 --iters 8 --check
  TVDN_VMM_PEER=0 TVDN_PEER_CHECK=1 TVDN_VMM_MIN_MIB=1 TVDN_GRANULE_MIB=2 step device_list_resident_plain 300 python3 tools/device_list_resident.py --shape 32x64x64x64 --devices 0,0,0 --iters 8 --check
elif [ "$NGPU" -ge 2 ]; then
  # config 5 in structure at a size every node can pin: the config-2 planes, 32 rows per rank, one GPU per rank
  step staged_slabs 1200 python3 tools/bench_staged_slabs.py --shape $((32 * NGPU))x256x128x128 --ranks "$NGPU" --gpu-per-rank --rows 8 --k 24 --iters 48
  step device_list_streamed 1200 python3 tools/device_list_streamed.py --shape $((32 * NGPU))x256x128x128 --devices "$(seq -s, 0 $((NGPU - 1)))" --rows 8 --k 24 --iters 48
  # resident slabs in one process: 32 rows of config-4 planes per GPU (62 GiB of state each), granules with peer access vs plain blocks
  step device_list_resident 900 python3 tools/device_list_resident.py --shape $((32 * NGPU))x512x256x256 --devices "$(seq -s, 0 $((NGPU - 1)))" --iters 20 --check
  TVDN_VMM_PEER=0 step device_list_resident_plain 900 python3 tools/device_list_resident.py --shape $((32 * NGPU))x512x256x256 --devices "$(seq -s, 0 $((NGPU - 1)))" --iters 20
fi
python3 tools/first_node_report.py "$O" | tee "$O/report.txt"
