# round 4: occupancy targets (amdgpu_waves_per_eu) of the fused sweep, variants alternating on one allocation (tools/ab_inproc.py)
set -o pipefail
O=gpurun_out/r4f; mkdir -p $O
U=tools/ubench
V() { for w in "$@"; do printf '%s ' "w$w:TVDN_LIB=$U/libtvdn_hip_wav$w.so"; done; }
for cfg in 3 plain32 2 3dplain f64fista; do
  timeout -k 10 240 python tools/ab_inproc.py --config $cfg --rounds 3 --steps 10 "base:" $(V 3 4 5 6 7 8) > $O/ab_waves_$cfg.jsonl 2> $O/ab_waves_$cfg.err || { echo "ab $cfg failed"; tail -3 $O/ab_waves_$cfg.err; }
  python - <<PY
import json
for l in open("$O/ab_waves_$cfg.jsonl"):
    d=json.loads(l); print(d["config"], d["variant"], d["mean_ms"], d["vs_first"], d["rounds"])
PY
done
export TVDN_STREAM_TIMING=1
timeout -k 10 200 python tools/stream_rates.py 64x1024x256x256 -1 -1 80 > $O/half_auto.json 2> $O/half_auto.err; grep "tvdn_run streamed" $O/half_auto.err
python -c "
import json;d=json.load(open('$O/half_auto.json'));print({k:d.get(k) for k in ('value','value_whole_call','stream_k','resident_rows','passes_s','setup_s','whole_call_s','h2d_GBps','d2h_GBps')})"
