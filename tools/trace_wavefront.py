"""Condense a rocprofv3 kernel + memory-copy trace of a wavefront pass: busy time per stream kind, copy sizes and rates,
and how much of the H2D / D2H time overlaps each other and the sweeps."""
import csv
import glob
import os
import sys
from collections import defaultdict


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def intervals_union(iv):
    iv = sorted(iv)
    out, cur = [], None
    for a, b in iv:
        if cur is None or a > cur[1]:
            if cur:
                out.append(cur)
            cur = [a, b]
        else:
            cur[1] = max(cur[1], b)
    if cur:
        out.append(cur)
    return out


def overlap(u1, u2):
    i = j = 0
    tot = 0
    while i < len(u1) and j < len(u2):
        a, b = max(u1[i][0], u2[j][0]), min(u1[i][1], u2[j][1])
        if a < b:
            tot += b - a
        if u1[i][1] < u2[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main(d):
    kt = newest(os.path.join(d, "**", "*kernel_trace.csv"))
    mt = newest(os.path.join(d, "**", "*memory_copy_trace.csv"))
    kern = defaultdict(list)
    for r in csv.DictReader(open(kt)):
        name = r["Kernel_Name"].split("(")[0][:60]
        kern[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    copies = defaultdict(list)
    for r in csv.DictReader(open(mt)):
        copies[r["Direction"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Size", 0) or 0)))
    t0 = min(min(a for a, _ in v) for v in kern.values())
    t1 = max(max(b for _, b in v) for v in kern.values())
    print(f"span {(t1 - t0) / 1e6:.1f} ms")
    allk = []
    for name, v in sorted(kern.items(), key=lambda kv: -sum(b - a for a, b in kv[1])):
        busy = sum(b - a for a, b in v)
        allk += v
        print(f"  kernel {name:60s} n={len(v):6d} busy={busy / 1e6:9.1f} ms avg={busy / len(v) / 1e3:8.1f} us")
    uk = intervals_union(allk)
    print(f"kernels union busy {sum(b - a for a, b in uk) / 1e6:.1f} ms")
    un = {}
    for dname, v in copies.items():
        busy = sum(b - a for a, b, _ in v)
        size = sum(s for _, _, s in v)
        big = [(a, b, s) for a, b, s in v if b - a > 100000]   # > 0.1 ms: the staging copies (the trace has no size column)
        un[dname] = intervals_union([(a, b) for a, b, _ in big])
        ub = sum(b - a for a, b in un[dname])
        rate = 0.0
        print(f"  copy {dname:20s} n={len(v):6d} (>=1MiB: {len(big)}) bytes={size / 2**30:8.2f} GiB busy={busy / 1e6:9.1f} ms "
              f"union={ub / 1e6:9.1f} ms  rate while copying={rate:6.2f} GB/s  avg size={size / max(1, len(v)) / 2**20:.1f} MiB")
    names = list(un)
    for i in range(len(names)):
        print(f"  {names[i]} overlapped with kernels: {overlap(un[names[i]], uk) / 1e6:.1f} ms")
        for j in range(i + 1, len(names)):
            print(f"  {names[i]} overlapped with {names[j]}: {overlap(un[names[i]], un[names[j]]) / 1e6:.1f} ms")
    # per-copy durations by size class
    for dname, v in copies.items():
        by = defaultdict(list)
        for a, b, s in v:
            by[s].append(b - a)
        for s, ds in sorted(by.items()):
            if len(ds) > 10:
                ds.sort()
                print(f"  {dname} size {s / 2**20:7.1f} MiB n={len(ds):5d} median {ds[len(ds) // 2] / 1e3:9.1f} us "
                      f"({s / ds[len(ds) // 2]:.1f} GB/s) p10 {ds[len(ds) // 10] / 1e3:.1f} p90 {ds[len(ds) * 9 // 10] / 1e3:.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
