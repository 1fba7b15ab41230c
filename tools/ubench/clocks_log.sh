#!/bin/bash
# logs the device's clocks, power, temperatures and throttle status (amd-smi, else rocm-smi) about twice a second until the file
# $1.stop appears (measurement aid; a separate process that makes no HIP call)
out=$1
if amd-smi metric -g 0 --json >/dev/null 2>&1; then
  while [ ! -e "$out.stop" ]; do
    echo "{\"t\": $(date +%s.%N), \"amdsmi\": $(amd-smi metric -g 0 --clock --power --temperature --perf-level --throttle --usage --json 2>/dev/null | tr -d '\n')}" >> "$out"
    sleep 0.3
  done
else
  while [ ! -e "$out.stop" ]; do
    echo "{\"t\": $(date +%s.%N), \"smi\": $(rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n')}" >> "$out"
    sleep 0.4
  done
fi
