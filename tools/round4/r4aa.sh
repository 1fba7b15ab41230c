set -o pipefail
O=gpurun_out/r4aa; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_run_streamed.py tests/test_gpu_outofcore.py tests/test_gpu_nonfinite_wrap.py tests/test_gpu_cubeio.py tests/test_gpu_two_ranks.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
TVDN_STREAM_TIMING=1 timeout -k 10 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
grep "tvdn_run streamed" $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4aa/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("best_placement"))
for e in d.get("also",[]):
    print({k:e.get(k) for k in ("value","value_whole_call","value_later_passes","first_pass_s","ms_per_step","stream_k","resident_rows","passes","passes_s","setup_s","whole_call_s","background_release_s","h2d_GBps","d2h_GBps","skipped","error") if e.get(k) is not None}, e["config"]["workload"][:60])
PY
