# whole suite as the driver runs it, then the default bench command with the library's streamed timing on stderr
set -o pipefail
O=gpurun_out/r4z; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/whole_suite.log 2>&1 || { tail -30 $O/whole_suite.log; exit 1; }
tail -2 $O/whole_suite.log
TVDN_STREAM_TIMING=1 timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
grep "tvdn_run streamed" $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4z/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("best_placement"))
for e in d.get("also",[]):
    print({k:e.get(k) for k in ("value","value_whole_call","value_later_passes","first_pass_s","ms_per_step","stream_k","resident_rows","passes","passes_s","setup_s","whole_call_s","h2d_GBps","d2h_GBps","skipped","error") if e.get(k) is not None}, e["config"]["workload"][:60])
print(d["cpu_baseline"])
PY
