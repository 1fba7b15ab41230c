"""One line per tools/bench_outofcore.py JSON on stdin: label rows k value seconds h2d d2h."""
import json
import sys
label = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        print(label, d["block_rows"], d["iters_per_pass"], d["value"], d["seconds"], d["h2d_GBps"], d["d2h_GBps"], flush=True)
