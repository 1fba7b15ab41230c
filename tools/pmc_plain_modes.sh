#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes for the forms the default bench line does not carry: the 4-D
# unaccelerated f32 sweep and the 3-D unaccelerated 512^3 sweep.  Counters in separate runs, the program itself after `--`.
R=$(pwd); O=$R/gpurun_out/r3pmc_plain; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "plain4d --plain" "plain3d --plain --shape 512x512x512"; do
  set -- $cfg; tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${tag}_$c -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-sustained --no-also "$@" > $O/${tag}_$c.json 2> $O/${tag}_$c.log || { echo "$tag $c failed"; tail -3 $O/${tag}_$c.log; exit 1; }
  done
  cd $R && python3 tools/pmc_summary.py $O/${tag}_FETCH_SIZE $O/${tag}_WRITE_SIZE $O/r03_c_$tag > $O/r03_c_pmc_${tag}_summary.txt 2>&1; cd /tmp
  grep "fused_iter" $O/r03_c_pmc_${tag}_summary.txt
  head -c 400 $O/${tag}_FETCH_SIZE.json; echo
done
