#!/usr/bin/env python3
"""One page from the logs of tools/first_node_run.sh: which steps ran, the JSON line of every bench.py run (value, transport,
pre-flight verdicts, ranks seen / distinct GPUs) and the weak-scaling efficiency against the 1-GPU line."""
import json
import os
import sys


def last_json(path):
    try:
        for line in reversed(open(path).read().splitlines()):
            line = line.strip()
            if line.startswith("{") and line.endswith("}"):
                return json.loads(line)
    except (OSError, ValueError):
        pass
    return None


def main():
    o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/first_node"
    print(open(os.path.join(o, "summary.txt")).read().rstrip())
    base = None
    def key(f):          # bench_gpus_2.out, bench_gpus_2_plain.out (TVDN_VMM=0: the state on plain hipMalloc memory)
        n, _, variant = f[len("bench_gpus_"):-4].partition("_")
        return int(n), variant

    for name in sorted((f for f in os.listdir(o) if f.startswith("bench_gpus_") and f.endswith(".out")), key=key):
        d = last_json(os.path.join(o, name))
        n, variant = key(name)
        if d is None:
            print(f"--gpus {n}{' ' + variant if variant else ''}: no JSON line (see {name[:-4]}.err)")
            continue
        pf = d.get("preflight") or {}
        per_gpu = d["value"] / max(1, d["n_gpus"])
        if n == 1:
            base = d["value"]
        eff = f", {d['value'] / base:.2f}x the 1-GPU line" if base else ""
        ok = (pf.get("ranks_seen") in (None, d["n_gpus"])) and (pf.get("distinct_gpus") in (None, d["n_gpus"]))
        print(f"--gpus {n}{' ' + variant if variant else ''}: {d['value']} {d['unit']} ({per_gpu:.1f} per GPU{eff}), {d['ms_per_step']} ms per step, "
              f"state on {d['config'].get('state_mem')}, "
              f"transport {d['config'].get('transport')}, overlap {d['config'].get('overlap')}, fallback {d.get('transport_fallback')}, "
              f"preflight overlap/blocking {pf.get('overlap')}/{pf.get('blocking')}, ranks seen {pf.get('ranks_seen')} of {pf.get('world')}, "
              f"distinct GPUs {pf.get('distinct_gpus')}{'' if ok else '  <-- NOT one GPU per rank'}, roofline frac {d['roofline']['frac']}")
    d = last_json(os.path.join(o, "staged_slabs.out"))
    if d:
        print(f"staged slabs (one process per GPU): {d.get('value')} {d.get('unit')} on {d.get('ranks')} ranks, shape {d.get('shape')}, "
              f"chunk rows {d.get('chunk_rows')}, k {d.get('k')}, whole call of every rank {d.get('seconds')} s")
    d = last_json(os.path.join(o, "device_list_streamed.out"))
    if d:
        print(f"streamed device list (one process): {d.get('value')} {d.get('unit')} passes only ({d.get('value_whole_call')} whole call) on "
              f"{d.get('distinct_devices')} GPUs, shape {d.get('shape')}, chunk rows {d.get('stream_rows')}, k {d.get('stream_k')}, "
              f"PCIe all devices {d.get('h2d_GBps_all')} + {d.get('d2h_GBps_all')} GB/s, bit-identical to one device: {d.get('bit_identical_to_one_device_resident')}")


    for name in ("device_list_resident", "device_list_resident_plain"):
        d = last_json(os.path.join(o, name + ".out"))
        if d:
            print(f"{name} (one process, peer copies): {d.get('value')} {d.get('unit')} on {d.get('distinct_devices')} GPUs, shape {d.get('shape')}, "
                  f"peer_check {d.get('peer_check')} (1 = a peer copy out of every granule block arrived intact, -1 = fell back to plain blocks, 0 = plain blocks), "
                  f"allocator faults {sum(m.get('faults', 0) for m in (d.get('mem') or {}).values())}, bit-identical to one device: {d.get('bit_identical_to_one_device')}")


if __name__ == "__main__":
    main()
