"""Segment-sharded execution: one process per GPU, segments partitioned across
ranks, one exchange step.

The reference has no distributed runtime; its harness simply loops
`for (auto& segment : reader)` into ONE heap (utils/index-search.cpp:719-779).
Here every rank executes the batch on its own segments, then all ranks
all-gather their per-segment top-k lists (RCCL over xGMI when the backend is
"nccl"; k*8 B per query per segment — latency-bound, so the whole batch goes in
one collective) and merge them on the GPU with irs_hip_merge_topk in the order
(score desc, segment asc, doc asc).

torch is plumbing here: device buffers for the collective and torch.distributed.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def segments_of_rank(n_segments: int, rank: int, world: int):
    """Contiguous block partition of segment ordinals (8 segments on 1/2/4/8 GPUs
    -> 8/4/2/1 per GPU)."""
    per = (n_segments + world - 1) // world
    return list(range(rank * per, min(n_segments, (rank + 1) * per)))


def _ptr_array(ptrs):
    return (C.c_void_p * len(ptrs))(*ptrs)


class TopkExchange:
    """The one exchange step of the path, with every buffer allocated once.

    Each rank owns `per` segment slots.  Its send buffer is ONE int64 tensor: `per` hit
    tables [nq][k] (an int64 carries one irs_hip_hit) followed by `per` count tables [nq]
    (int32, two per int64) — `slot(i)` hands out the device pointers of slot i so that
    irs_hip_batch_results_to_device writes straight into it.  `run()` = one
    all_gather_into_tensor (RCCL over xGMI with backend "nccl") + irs_hip_merge_topk.
    """

    def __init__(self, L, device_index: int, n_segments: int, rank: int, world: int, nq: int,
                 k: int, tensor_device):
        self.L, self.device_index, self.nq, self.k = L, device_index, nq, k
        self.world, self.rank = world, rank
        self.per = per = (n_segments + world - 1) // world
        self.n_lists = min(n_segments, world * per)
        self.hit_words = per * nq * k
        self.cnt_words = (per * nq + 1) // 2
        words = self.hit_words + self.cnt_words
        self.send = torch.zeros((words,), dtype=torch.int64, device=tensor_device)
        self.recv = (torch.zeros((world * words,), dtype=torch.int64, device=tensor_device)
                     if world > 1 else self.send)
        self.out_h = torch.zeros((nq, k), dtype=torch.int64, device=tensor_device)
        self.out_s = torch.zeros((nq, k), dtype=torch.int32, device=tensor_device)
        self.out_c = torch.zeros((nq,), dtype=torch.int32, device=tensor_device)
        base = self.recv.data_ptr()
        lists, counts = [], []
        for i in range(self.n_lists):       # global segment ordinal i = rank r, slot j
            r, j = divmod(i, per)
            blk = base + 8 * r * words
            lists.append(blk + 8 * j * nq * k)
            counts.append(blk + 8 * self.hit_words + 4 * j * nq)
        self._lists, self._counts = _ptr_array(lists), _ptr_array(counts)
        self._seg_ids = np.arange(self.n_lists, dtype=np.uint32)

    def slot(self, i: int):
        """(hits pointer, counts pointer) of local segment slot i in the send buffer."""
        base = self.send.data_ptr()
        return base + 8 * i * self.nq * self.k, base + 8 * self.hit_words + 4 * i * self.nq

    def gather(self, async_op: bool = False):
        """The collective alone; with async_op the work handle (None on a single rank)."""
        if self.world > 1:
            return dist.all_gather_into_tensor(self.recv, self.send, async_op=async_op)
        return None

    def merge(self, stream=None):
        _lib.check(self.L, self.L.irs_hip_merge_topk(
            self.device_index, self._lists, self._counts, self._seg_ids.ctypes.data,
            self.n_lists, self.nq, self.k, self.out_h.data_ptr(), self.out_s.data_ptr(),
            self.out_c.data_ptr(), stream), "irs_hip_merge_topk")
        return self.out_h, self.out_s, self.out_c

    def run(self, stream=None):
        if self.world > 1:
            dist.all_gather_into_tensor(self.recv, self.send)
        _lib.check(self.L, self.L.irs_hip_merge_topk(
            self.device_index, self._lists, self._counts, self._seg_ids.ctypes.data,
            self.n_lists, self.nq, self.k, self.out_h.data_ptr(), self.out_s.data_ptr(),
            self.out_c.data_ptr(), stream), "irs_hip_merge_topk")
        return self.out_h, self.out_s, self.out_c


class PipelinedExchange:
    """Two TopkExchange buffer sets used alternately, so that the all-gather of step i runs on
    the collective's own stream (RCCL) WHILE the kernels of step i+1 execute; the merge of
    step i is enqueued behind those kernels.  Per step:  run the batch (async) ->
    finish() = merge of the previous step -> results_to_device into slot(phase, .) ->
    start(phase).  After the last step: finish().  Every step still ends in a checked,
    merged, device-resident top-k — one step later."""

    def __init__(self, *args, **kw):
        self.ex = [TopkExchange(*args, **kw), TopkExchange(*args, **kw)]
        self.pending = None

    def slot(self, phase: int, i: int):
        return self.ex[phase].slot(i)

    def start(self, phase: int):
        assert self.pending is None
        self.pending = (phase, self.ex[phase].gather(async_op=True))

    def finish(self, stream=None):
        if self.pending is None:
            return None
        phase, work = self.pending
        self.pending = None
        if work is not None:
            work.wait()     # nccl: the current stream waits for the collective; gloo: the host
        return self.ex[phase].merge(stream)


def gather_merge(L, device_index: int, local_lists, n_segments: int, rank: int, world: int,
                 nq: int, k: int, tensor_device, stream=None):
    """local_lists: [(segment ordinal, hits tensor int64 [nq, k], counts tensor int32 [nq])]
    for this rank's segments.  Returns tensors (hits int64 [nq, k], seg int32 [nq, k],
    counts int32 [nq]) identical on all ranks.  (One-shot form of TopkExchange.)"""
    ex = TopkExchange(L, device_index, n_segments, rank, world, nq, k, tensor_device)
    cnt32 = ex.send[ex.hit_words:].view(torch.int32)
    for i, (_, h, c) in enumerate(local_lists):
        ex.send[i * nq * k:(i + 1) * nq * k].copy_(h.reshape(-1))
        cnt32[i * nq:(i + 1) * nq].copy_(c)
    return ex.run(stream)


def hits_from_int64(t: torch.Tensor) -> np.ndarray:
    return t.cpu().numpy().view(_lib.HIT).reshape(t.shape)
