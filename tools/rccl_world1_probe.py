#!/usr/bin/env python3
"""The RCCL bring-up of bench.py / denoise_slabs on a ONE-rank group (all a one-GPU box can do): gloo control group, RCCL data
group created with the same options (high-priority stream, device_id), a barrier, an all-reduce on device memory, and the
exchange self-check on the live group (world 1: no messages, but every call of the path runs).  Catches API drift in
torch.distributed before a node does.   torchrun --nproc-per-node 1 --master-addr 127.0.0.1 tools/rccl_world1_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

import bench


def main():
    out = {}
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    group, transport, fallback = bench.init_groups(local_rank)
    out.update(transport=transport, fallback=fallback, group_backend=dist.get_backend(group) if group is not None else None)
    t = torch.ones(4, device=f"cuda:{local_rank}")
    dist.all_reduce(t, group=group)
    torch.cuda.synchronize()
    out["all_reduce_ok"] = bool(t.sum().item() == 4.0)
    # Round 5: big states live on granules of HIP virtual memory (csrc/tvdn_devmem.hip).  RCCL reads and writes them as user
    # buffers: an all-reduce in place, and a send to oneself (one rank is all this box has: RCCL copies send -> receive buffer
    # with its own kernel), each on a block of granules, checked against the values put there.
    from cytvdn_amd import _lib
    blk = _lib.DeviceBlock(3 << 30, local_rank)          # 3 GiB: granules (the threshold is 2 GiB), crosses granule borders
    out["block_kind"] = {_lib.MEM_GRANULES: "granules", _lib.MEM_PLAIN: "plain"}.get(blk.kind, blk.kind)
    g = blk.tensor(torch.float32)
    plain = torch.empty(g.numel(), dtype=torch.float32, device=g.device)      # the control: torch's own memory, same sizes
    me = dist.get_rank(group)
    peer = me if group is None else dist.get_global_rank(group, me)
    for name, mem in (("plain", plain), ("granules", g)):
        # 256 MiB messages (a row-plane of BASELINE configs[4]) placed so that both straddle a granule border
        half = (256 << 20) // 4
        snd, rcv = mem[(1 << 28) - half // 2:(1 << 28) + half // 2], mem[(1 << 29) - half // 2:(1 << 29) + half // 2]
        snd.copy_((torch.arange(half, device=mem.device) % 4093).float())
        want = snd.clone()
        rcv.zero_()
        dist.all_reduce(snd, group=group)
        torch.cuda.synchronize()
        out[f"all_reduce_on_{name}_ok"] = bool(torch.equal(snd, want))
        try:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, snd, peer, group), dist.P2POp(dist.irecv, rcv, peer, group)]):
                w.wait()
            torch.cuda.synchronize()
            out[f"send_to_self_on_{name}_ok"] = bool(torch.equal(want, rcv))
            if not out[f"send_to_self_on_{name}_ok"]:
                out[f"send_to_self_on_{name}_differs"] = int((want != rcv).sum())
        except Exception as e:                          # a transport that refuses a send to oneself is not a failure of the memory
            out[f"send_to_self_on_{name}_ok"] = f"not run: {type(e).__name__}: {e}"
        del snd, rcv, want
    del g, plain, mem
    blk.free()
    torch.cuda.empty_cache()
    from cytvdn_amd.distributed import selfcheck_exchange
    out["preflight"] = selfcheck_exchange(group=group, device=local_rank)
    # the measurement path itself with the RCCL group handed in (world 1: SlabRunner skips the exchange)
    r = bench.measure((16, 64, 64, 64), "f32", True, "compact", 4, 1, local_rank, 0, 1, group)
    out["measure_value"] = r["value"]
    print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
