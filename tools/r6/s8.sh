#!/bin/bash
set -u -o pipefail
O=gpurun_out/r6; mkdir -p $O
timeout -k 10 600 python tools/ubench/one_row_probe.py > $O/s8_one_row_probe.jsonl 2> $O/s8.err; echo rc $?; cat $O/s8_one_row_probe.jsonl; tail -3 $O/s8.err
