# round 4: two ranks (processes, gloo) sharing ONE GPU, each streaming its half of 64x1024x256x256 with the library's loop
# (denoise_slabs(staged=...)): interior rows resident in HBM against every row streamed.  Each rank may count on 130 GB of HBM.
set -o pipefail
O=gpurun_out/r4at; mkdir -p $O
export TVDN_HBM_LIMIT=130G
run() { name=$1; shift; timeout -k 10 400 python tools/bench_staged_slabs.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -8 $O/$name.err; return 1; }; tail -1 $O/$name.json; }
run resident_all --shape 64x1024x256x256 --ranks 2 --rows 2 --k 8 --iters 32 --resident -1 &&
run resident_none --shape 64x1024x256x256 --ranks 2 --rows 2 --k 8 --iters 32 --resident 0 &&
run resident_none_k16 --shape 64x1024x256x256 --ranks 2 --rows 2 --k 16 --iters 32 --resident 0
