import os, sys
sys.path.insert(0, os.getcwd())
os.environ["TVDN_VMM_MIN_MIB"] = "1"
os.environ["TVDN_GRANULE_MIB"] = sys.argv[1] if len(sys.argv) > 1 else "8"
import numpy as np, torch
from cytvdn_amd import _lib
for nbytes in (8 << 20, (8 << 20) + 16, (40 << 20) - 4096, (17 << 20) + 4):
    b = _lib.DeviceBlock(nbytes, 0)
    t = b.tensor(torch.uint8)
    t.fill_(3)
    torch.cuda.synchronize()
    h = t.cpu().numpy()
    bad = np.nonzero(h != 3)[0]
    print(nbytes, "kind", b.kind, "ptr", hex(b.ptr), "t[0]", int(t[0]), "t[-1]", int(t[-1]), "mid", int(t[nbytes // 2]), "sum ok", int(t.sum(dtype=torch.int64)) == 3 * nbytes,
          "full D2H bad count", bad.size, "first bad", (int(bad[0]), int(bad[-1])) if bad.size else None, flush=True)
    # element reads via a kernel (gather) instead of the runtime's copy
    idx = torch.tensor([0, nbytes // 2, nbytes - 1], device="cuda")
    print("   gathered by a kernel:", t[idx].cpu().tolist(), flush=True)
    del t
    b.free()
