"""Condense the TVDN_WF_TRACE=1 timeline (stderr of a wavefront run): per phase the median duration, how many were
more than twice the median, and the chunk period in the middle of the pass."""
import sys
from collections import defaultdict
d = defaultdict(list)
for line in open(sys.argv[1]):
    if line.startswith("wf-trace"):
        f = line.split()
        d[f[1]].append((int(f[3]), float(f[4]), float(f[6])))
for kind, v in d.items():
    v.sort()
    mid = [b - a for c, a, b in v[len(v) // 4: 3 * len(v) // 4]]
    mid.sort()
    med = mid[len(mid) // 2]
    slow = sum(1 for c, a, b in v if b - a > 2 * med)
    starts = [a for c, a, b in v[len(v) // 4: 3 * len(v) // 4]]
    period = (starts[-1] - starts[0]) / max(1, len(starts) - 1)
    print(f"{kind:6s} n={len(v):4d} median {med:6.1f} ms  >2x median: {slow:3d}  max {max(b - a for c, a, b in v):6.1f}  period {period:6.1f} ms")
