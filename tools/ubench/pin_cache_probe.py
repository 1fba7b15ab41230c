#!/usr/bin/env python3
"""GPU box, harmless: does the runtime keep a caller's PAGEABLE array pinned after a host <-> device copy?  (The mechanism behind the
GPU memory-access fault of profiles/r06_abort_found.txt: the runtime pins pageable memory in place for copies above a few KiB and
caches the pinned object by address and size; memory freed, unmapped and mapped anew behind such an entry is a fault waiting for
the next copy from that address.)  hipPointerGetAttributes on the array before / after a copy by the runtime (torch's `copy_` =
hipMemcpy from pageable memory) and after a copy through the library's pinned lanes (tvdn_copy_to_device).  Nothing is freed
under the runtime's feet here: no fault is provoked."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cytvdn_amd import _lib


class Attr(C.Structure):
    _fields_ = [("type", C.c_int), ("device", C.c_int), ("devicePointer", C.c_void_p), ("hostPointer", C.c_void_p), ("isManaged", C.c_int),
                ("allocationFlags", C.c_uint)]


hip = None
for line in open("/proc/self/maps"):
    if "libamdhip64" in line:
        hip = C.CDLL(line.split()[-1])
        break
hip.hipPointerGetAttributes.argtypes = [C.POINTER(Attr), C.c_void_p]
hip.hipGetLastError.restype = C.c_int


def known(ptr):
    a = Attr()
    rc = hip.hipPointerGetAttributes(C.byref(a), C.c_void_p(ptr))
    hip.hipGetLastError()
    return {"rc": rc, "type": a.type if rc == 0 else None}      # rc 0 + type 1 (hipMemoryTypeHost): the runtime holds this range


_lib.ctx(0)
for nbytes in (2048, 64 << 10, 2600 << 10, 40 << 20):
    dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    x_rt = np.full(nbytes, 7, np.uint8)
    x_lib = np.full(nbytes, 9, np.uint8)
    out = {"bytes": nbytes, "heap_or_mmap": "heap" if x_rt.ctypes.data < (1 << 46) else "mmap", "before": known(x_rt.ctypes.data)}
    dev.copy_(torch.from_numpy(x_rt))
    torch.cuda.synchronize()
    out["after_runtime_copy_h2d"] = known(x_rt.ctypes.data)
    back = dev.cpu()
    out["after_runtime_copy_d2h_fresh_tensor"] = known(back.data_ptr())
    _lib.copy_to_device(x_lib, dev)
    out["after_library_copy_h2d"] = known(x_lib.ctypes.data)
    y = _lib.copy_to_host(dev, np.uint8)
    out["after_library_copy_d2h"] = known(y.ctypes.data)
    assert int(y[0]) == 9 and int(back[0]) == 7
    print(json.dumps(out), flush=True)
