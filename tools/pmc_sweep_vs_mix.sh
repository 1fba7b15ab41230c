#!/bin/bash
# PMC passes over tools/ceiling_vs_sweep.py (one placement, few steps): L2 hit/miss, memory-side requests and SQ wait
# buckets of the fused sweep next to the pure stream mix on the same arrays.  Counters only (no trace domains besides
# --kernel-trace); the program itself follows `--`.
R=$(pwd); O=$R/gpurun_out/r3pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
CFG=${1:-2}
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-60)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -- python3 $R/tools/ceiling_vs_sweep.py --config $CFG --hold 1 --steps 3 > $O/$tag.out 2> $O/$tag.log
  echo "$set rc=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.getcwd(), "gpurun_out", "r3pmc")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        n = r["Kernel_Name"]
        if "fused_iter_kernel" in n or "stream_mix_kernel" in n:
            k = "sweep" if "fused" in n else "mix"
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(O, "r03_pmc_sweep_vs_mix.txt"), "w") as fh:
    names = sorted(set(acc["sweep"]) | set(acc["mix"]))
    fh.write(f"{'counter':34s} {'sweep (mean/launch)':>22s} {'mix (mean/launch)':>22s} {'sweep/mix':>10s}\n")
    for c in names:
        s = sum(acc["sweep"][c]) / max(1, len(acc["sweep"][c])); m = sum(acc["mix"][c]) / max(1, len(acc["mix"][c]))
        fh.write(f"{c:34s} {s:22.0f} {m:22.0f} {(s / m if m else float('nan')):10.3f}\n")
print(open(os.path.join(O, "r03_pmc_sweep_vs_mix.txt")).read())
PY
