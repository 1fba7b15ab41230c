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
