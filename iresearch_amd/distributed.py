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


def gather_merge(L, device_index: int, local_lists, n_segments: int, rank: int, world: int,
                 nq: int, k: int, tensor_device, stream=None):
    """local_lists: [(segment ordinal, hits tensor int64 [nq, k], counts tensor int32 [nq])]
    for this rank's segments (an int64 carries one irs_hip_hit).  Returns tensors
    (hits int64 [nq, k], seg int32 [nq, k], counts int32 [nq]) identical on all ranks."""
    per = (n_segments + world - 1) // world
    send_h = torch.zeros((per, nq, k), dtype=torch.int64, device=tensor_device)
    send_c = torch.zeros((per, nq), dtype=torch.int32, device=tensor_device)
    for i, (_, h, c) in enumerate(local_lists):
        send_h[i].copy_(h)
        send_c[i].copy_(c)
    if world > 1:
        all_h = torch.empty((world * per, nq, k), dtype=torch.int64, device=tensor_device)
        all_c = torch.empty((world * per, nq), dtype=torch.int32, device=tensor_device)
        dist.all_gather_into_tensor(all_h, send_h)
        dist.all_gather_into_tensor(all_c, send_c)
    else:
        all_h, all_c = send_h, send_c
    n_lists = min(n_segments, world * per)
    seg_ids = np.arange(n_lists, dtype=np.uint32)
    out_h = torch.empty((nq, k), dtype=torch.int64, device=tensor_device)
    out_s = torch.empty((nq, k), dtype=torch.int32, device=tensor_device)
    out_c = torch.empty((nq,), dtype=torch.int32, device=tensor_device)
    lists = _ptr_array([all_h[i].data_ptr() for i in range(n_lists)])
    counts = _ptr_array([all_c[i].data_ptr() for i in range(n_lists)])
    _lib.check(L, L.irs_hip_merge_topk(device_index, lists, counts, seg_ids.ctypes.data, n_lists,
                                       nq, k, out_h.data_ptr(), out_s.data_ptr(),
                                       out_c.data_ptr(), stream), "irs_hip_merge_topk")
    return out_h, out_s, out_c


def hits_from_int64(t: torch.Tensor) -> np.ndarray:
    return t.cpu().numpy().view(_lib.HIT).reshape(t.shape)
