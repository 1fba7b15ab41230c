#!/usr/bin/env python3
"""Interleaved A/B timing on one GPU box: runs `bench.py` (or tools/time_kernel_level.py with --passes) once per variant
per round, alternating the variants, and prints mean / min of the sweep-kernel time per variant.  Boxes and moments differ
by several per cent, so only numbers taken alternately on one box in one call are comparable.

    python tools/ab.py --rounds 4 "base:" "skew4k:TVDN_ARRAY_SKEW=4096" -- --steps 20 --warmup 3
    python tools/ab.py --passes --rounds 3 "c8:TVDN_PASS_CHUNK=8" "c32:TVDN_PASS_CHUNK=32;TVDN_LIB=/path/variant.so"
"""
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    rounds, passes = 3, False
    if "--rounds" in args:
        i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
    if "--passes" in args:
        passes = True; args.remove("--passes")
    extra = []
    if "--" in args:
        i = args.index("--"); extra = args[i + 1:]; args = args[:i]
    variants = []
    for v in args:
        label, _, envs = v.partition(":")
        env = dict(e.split("=", 1) for e in envs.split(";") if e)   # several: "A=1;B=2" (values may hold commas)
        variants.append((label, env))
    res = defaultdict(lambda: defaultdict(list))
    for r in range(rounds):
        for label, env in variants:
            e = dict(os.environ); e.update(env)
            if passes:
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_kernel_level.py")] + extra, env=e,
                                     capture_output=True, text=True).stdout
                for line in out.splitlines():
                    if line.startswith("{"):
                        d = json.loads(line)
                        res[label][f"{d['op']} {d['dtype']}"].append(d["ms"])
            else:
                out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-also"] + extra,
                                     env=e, capture_output=True, text=True).stdout
                line = [l for l in out.splitlines() if l.startswith("{")]
                if line:
                    d = json.loads(line[0])
                    res[label]["kernel_ms"].append(d["roofline"]["kernel_ms"])
                    res[label]["value"].append(d["value"])
    for label, _ in variants:
        for k, v in res[label].items():
            print(f"{label:14s} {k:40s} mean {sum(v) / len(v):9.4f}  min {min(v):9.4f}  n={len(v)}  all={[round(x, 3) for x in v]}", flush=True)


if __name__ == "__main__":
    main()
