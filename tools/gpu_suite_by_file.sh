#!/usr/bin/env bash
# Runs the `-m gpu` tests FILE BY FILE, each file in its own pytest process under its own timeout, logs kept per file
# under gpurun_out/<tag>/ (merged back by gpurun), and stops at the first file that fails or times out -- after a GPU
# step has been killed no further GPU step is started in the same call.
#   tools/gpu_suite_by_file.sh <tag> <per-file-timeout-s> tests/test_gpu_a.py tests/test_gpu_b.py ...
# Round 2 lost two boxes to one test inside a whole-suite run and kept no log of the last good suite; this is the
# procedure VERDICT r2 asked for instead (cheapest files first, the misfit file last, one summary line per file).
set -u -o pipefail
tag=$1; tmo=$2; shift 2
out=gpurun_out/$tag
mkdir -p "$out"
export TVDN_HOST_LIMIT=${TVDN_HOST_LIMIT:-64G}
summary=$out/summary.txt
echo "commit $(cat .git_head 2>/dev/null || echo unknown)  $(date -u +%FT%TZ)  $(hostname)" >> "$summary"
for f in "$@"; do
    name=$(basename "$f" .py)
    echo "=== $name" | tee -a "$summary"
    timeout -k 10 "$tmo" python -m pytest "$f" -m gpu -x -q -p no:cacheprovider --timeout 600 > "$out/$name.log" 2>&1
    rc=$?
    tail -n 1 "$out/$name.log" | tee -a "$summary"
    echo "rc=$rc" | tee -a "$summary"
    if [ $rc -ne 0 ]; then
        tail -n 40 "$out/$name.log"
        exit $rc
    fi
done
echo "ALL FILES GREEN" | tee -a "$summary"
