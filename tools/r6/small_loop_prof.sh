#!/bin/bash
# kernel time against iteration time of small-cube loops (rocprofv3 --kernel-trace --stats), one line per kernel
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for a in "64x64x256 2000" "64x64x256 2000 rule" "128x128x512 1000" "128x128x512 1000 rule"; do
  rm -rf /tmp/p
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -- python3 $R/tools/r6/small_loop.py $a 2>/dev/null | grep "us per"
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/p/**/*kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0])))[:4]: print("   ", r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
done
