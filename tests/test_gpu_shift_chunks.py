"""The device-staged form of the K-row swap between ranks (`distributed._RankHooks._shift` as it runs over RCCL: rows staged
through two reusable device buffers, array by array, in chunks of at most STAGE_BYTES) on ONE GPU: the ranks are threads, and
a stand-in for torch.distributed pairs their batched sends and receives through queues.  What RCCL itself does is not covered
here (tests/test_gpu_rccl.py needs the GPUs); what is: both sides cut the same chunks, tags and peers pair up in a chain and in
a ring, ragged last chunks, and every halo row arrives where the library expects it."""
import queue
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Wire:
    """Mailboxes shared by the ranks: (source, destination) -> queue of (tag, rows)."""

    def __init__(self, world):
        self.box = {(s, d): queue.Queue() for s in range(world) for d in range(world)}
        self.sent_bytes = [0] * world
        self.largest = 0


class _Dist:
    """As much of torch.distributed as _RankHooks uses, for one rank."""
    isend, irecv = "isend", "irecv"

    class P2POp:
        def __init__(self, op, tensor, peer, group=None, tag=0):
            self.op, self.tensor, self.peer, self.tag = op, tensor, peer, tag

    class _Done:
        def wait(self):
            return True

    def __init__(self, wire, rank):
        self.wire, self.rank = wire, rank

    def get_backend(self, group=None):
        return "nccl"

    def batch_isend_irecv(self, ops):
        assert ops, "an empty batch is an error in torch.distributed"
        for o in ops:
            if o.op == self.isend:
                assert o.tensor.is_cuda and o.tensor.is_contiguous()
                self.wire.sent_bytes[self.rank] += o.tensor.numel()
                self.wire.largest = max(self.wire.largest, o.tensor.numel())
                self.wire.box[(self.rank, o.peer)].put((o.tag, o.tensor.clone()))
        for o in ops:
            if o.op == self.irecv:
                assert o.tensor.is_cuda
                tag, rows = self.wire.box[(o.peer, self.rank)].get(timeout=30)
                assert tag == o.tag and rows.shape == o.tensor.shape, (tag, o.tag, rows.shape, o.tensor.shape)
                o.tensor.copy_(rows)
        return [self._Done() for _ in ops]


@pytest.mark.parametrize("world,periodic,depth,stage_rows", [
    (2, False, 7, 3),      # chain, chunks of 3 + 3 + 1 rows
    (3, True, 5, 2),       # ring: every rank sends and receives both ways
    (2, True, 3, 2),       # ring of two: the same neighbour on both sides
    (3, False, 4, 100),    # one chunk holds the whole swap
    (4, True, 3, 1),       # row by row
])
def test_device_staged_swap_pairs_up_in_chunks(monkeypatch, world, periodic, depth, stage_rows):
    import torch
    from cytvdn_amd.distributed import _RankHooks
    row_bytes, own, n_arrays = 4096, 9, 5
    monkeypatch.setattr(_RankHooks, "STAGE_BYTES", stage_rows * row_bytes)
    rows_per = own + 2 * depth
    lo, hi = depth, depth + own
    rng = np.random.default_rng(5)
    arrays = [[torch.from_numpy(rng.integers(0, 255, (rows_per, row_bytes), dtype=np.uint8)) for _ in range(n_arrays)] for _ in range(world)]
    before = [[t.clone() for t in a] for a in arrays]
    wire = _Wire(world)
    errors = []

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            h = _RankHooks(_Dist(wire, r), None, r, world, 0, periodic)
            assert h.via_dev
            # the two shifts of _RankHooks.exchange
            h._shift(arrays[r], slice(hi - depth, hi), slice(lo - depth, lo), h.right, h.left, 2)
            h._shift(arrays[r], slice(lo, lo + depth), slice(hi, hi + depth), h.left, h.right, 1)
            assert h._stage[0].shape[0] == min(depth, stage_rows)            # two buffers of one chunk, whatever the swap's size
        except BaseException as e:
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert not errors, errors
    assert wire.largest <= max(1, min(depth, stage_rows)) * row_bytes
    for r in range(world):
        left = (r - 1) % world if (r > 0 or periodic) else None
        right = (r + 1) % world if (r < world - 1 or periodic) else None
        for i in range(n_arrays):
            assert torch.equal(arrays[r][i][lo:hi], before[r][i][lo:hi])                     # own rows untouched
            want_lo = before[left][i][hi - depth:hi] if left is not None else before[r][i][:lo]
            want_hi = before[right][i][lo:lo + depth] if right is not None else before[r][i][hi:]
            assert torch.equal(arrays[r][i][:lo], want_lo), (r, i, "low halo")
            assert torch.equal(arrays[r][i][hi:], want_hi), (r, i, "high halo")
    for q in wire.box.values():
        assert q.empty()
