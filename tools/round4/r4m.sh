set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r4m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export TVDN_STREAM_TIMING=1
for mode in 1 0; do
  export TVDN_STREAM_CHAIN=$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_chain$mode -- python3 $GRAFT_REPO_ROOT/tools/stream_rates.py 32x1024x256x256 2 8 24 0 > $O/chain$mode.json 2> $O/chain$mode.err
  grep "tvdn_run streamed: rows" $O/chain$mode.err | cut -c1-160
  python3 $GRAFT_REPO_ROOT/tools/trace_wavefront.py $O/trace_chain$mode > $O/trace_chain${mode}_summary.txt 2>&1; cat $O/trace_chain${mode}_summary.txt
done
