"""Does the D2H rate into a large pinned array depend on how its pages are backed?  Compares torch's pinned allocation
with an mmap'ed, MADV_HUGEPAGE'd, first-touched and hipHostRegister'ed one: bursts of 9 x 64 MiB D2H copies walking
through 8 arrays of 4 GiB (the wavefront engine's pattern), while H2D copies run beside them."""
import ctypes
import mmap
import sys

import torch

dev = torch.device("cuda", 0)
MiB, GiB = 1 << 20, 1 << 30
N_ARR, ARR = 9, 4 * GiB
PIECE = 64 * MiB
rt = torch.cuda.cudart()


def smaps_huge(addr):
    """AnonHugePages / Size of the mapping containing addr."""
    size = huge = None
    hit = False
    for line in open("/proc/self/smaps"):
        f = line.split()
        if "-" in f[0] and len(f) >= 5 and all(c in "0123456789abcdef-" for c in f[0]):
            a, b = (int(x, 16) for x in f[0].split("-"))
            hit = a <= addr < b
        elif hit and f[0] == "Size:":
            size = int(f[1])
        elif hit and f[0] == "AnonHugePages:":
            huge = int(f[1])
            return size, huge
    return size, huge


def alloc_torch():
    t = torch.empty(ARR, dtype=torch.uint8, pin_memory=True)
    t.zero_()
    return t, None


def alloc_thp():
    m = mmap.mmap(-1, ARR + 2 * MiB, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    m.madvise(mmap.MADV_HUGEPAGE)
    buf = (ctypes.c_char * (ARR + 2 * MiB)).from_buffer(m)
    base = ctypes.addressof(buf)
    off = (-base) % (2 * MiB)
    t = torch.frombuffer(m, dtype=torch.uint8, count=ARR, offset=off)
    t.zero_()
    err = rt.cudaHostRegister(t.data_ptr(), ARR, 0)
    assert int(err) == 0, err
    return t, m


dbuf = torch.empty(N_ARR * PIECE, dtype=torch.uint8, device=dev)
ubuf = torch.empty(10 * PIECE, dtype=torch.uint8, device=dev)
uhost = torch.empty(10 * PIECE, dtype=torch.uint8, pin_memory=True)
s_down, s_up = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

for name, alloc in (("torch-pinned", alloc_torch), ("thp-registered", alloc_thp)):
    arrs = [alloc() for _ in range(N_ARR)]
    sz, hg = smaps_huge(arrs[0][0].data_ptr())
    print(f"{name}: mapping of array 0: Size {sz} kB AnonHugePages {hg} kB", flush=True)
    times = []
    for step in range(ARR // PIECE):
        torch.cuda.synchronize()
        with torch.cuda.stream(s_up):
            ubuf.copy_(uhost, non_blocking=True)
        with torch.cuda.stream(s_down):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(s_down)
            for i, (t, _) in enumerate(arrs):
                t[step * PIECE:(step + 1) * PIECE].copy_(dbuf[i * PIECE:(i + 1) * PIECE], non_blocking=True)
            e1.record(s_down)
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    srt = sorted(times)
    print(f"{name}: {len(times)} bursts of 9 x 64 MiB: min {srt[0]:.1f} median {srt[len(srt) // 2]:.1f} max {srt[-1]:.1f} ms; "
          f"bursts slower than 2x min: {sum(1 for t in times if t > 2 * srt[0])}", flush=True)
    print("   per burst:", " ".join(f"{t:.0f}" for t in times), flush=True)
    for t, m in arrs:
        if m is not None:
            rt.cudaHostUnregister(t.data_ptr())
    del arrs
