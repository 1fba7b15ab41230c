#!/usr/bin/env python3
"""Samples the GPU's clocks / power / temperature once per second (sysfs first, `rocm-smi` as a fallback) while a command
runs, so that the two states a box shows on the config-2 sweep (11.2 ms vs 12.5 ms) can be tied to what
the device reports.  The command is started as a CHILD (never exec'ed over this process); this process never touches HIP.

    python tools/clocks_during.py gpurun_out/clocks.txt -- python bench.py --steps 600 --no-also --no-cpu-baseline
"""
import glob
import json
import os
import subprocess
import sys
import time


def sysfs_sources():
    """Every amdgpu card the box shows (the container sees all 8 of the host; which one is ours shows in gpu_busy_percent)."""
    src = {}
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if not os.path.exists(os.path.join(card, "pp_dpm_sclk")):
            continue
        c = os.path.basename(os.path.dirname(card))
        for name in ("gpu_busy_percent", "mem_busy_percent"):
            p = os.path.join(card, name)
            if os.path.exists(p):
                src[f"{c}.{name}"] = p
        for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            for f in sorted(os.listdir(hw)):
                if f.startswith(("freq", "power", "temp")) and f.endswith(("_input", "_average")):
                    src[f"{c}.{f}"] = os.path.join(hw, f)
    return src


def read(path):
    try:
        t = open(path).read().strip()
    except OSError as e:
        return "ERR:" + e.__class__.__name__
    if "\n" in t:                      # pp_dpm_*: the level marked with '*'
        cur = [l for l in t.splitlines() if l.rstrip().endswith("*")]
        return cur[0].split(":")[1].strip(" *") if cur else t.replace("\n", "|")
    return t


def smi_sample():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showuse", "--json"],
                             capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = next(iter(d.values()))
        keep = {k: v for k, v in card.items() if any(s in k.lower() for s in ("sclk", "mclk", "fclk", "power", "temperature", "use"))}
        return keep
    except Exception as e:
        return {"rocm-smi": repr(e)}


def main():
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    src = sysfs_sources()
    use_smi = not src or os.environ.get("CLOCKS_SMI") == "1"
    with open(out_path, "w") as f:
        f.write("# sources: %s\n" % (json.dumps(src) if not use_smi else "rocm-smi --showclocks --showpower --showtemp --showuse"))
        f.write("# command: %s\n" % " ".join(cmd))
        f.flush()
        child = subprocess.Popen(cmd)
        t0 = time.time()
        while True:
            row = {"t": round(time.time() - t0, 2)}
            if use_smi:
                row.update(smi_sample())
            else:
                row.update({k: read(p) for k, p in src.items()})
            f.write(json.dumps(row) + "\n")
            f.flush()
            if child.poll() is not None:
                break
            time.sleep(max(0.0, 1.0 - ((time.time() - t0) % 1.0)))
        f.write("# exit %d after %.1f s\n" % (child.returncode, time.time() - t0))
    sys.exit(child.returncode)


if __name__ == "__main__":
    main()
