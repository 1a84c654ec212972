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


class Communicator:
    """irs_hip_comm: the RCCL communicator behind the C ABI (include/irs_hip.h) — the data path
    of the exchange does not go through torch.distributed.  The 128-byte id is made on rank 0
    and handed to the other ranks by `broadcast` (any callable: here torch.distributed's object
    broadcast, in a C++ host MPI or a file)."""

    def __init__(self, L, device_index: int, rank: int, world: int, broadcast=None):
        self.L, self.rank, self.world = L, rank, world
        ident = (C.c_uint8 * 128)()
        if rank == 0 and (broadcast is None or world == 1):
            _lib.check(L, L.irs_hip_comm_unique_id(ident), "irs_hip_comm_unique_id")
        if world > 1:
            if broadcast is None:
                def broadcast(b):
                    box = [b]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
            # (a caller-supplied `broadcast` returns the id on every rank, rank 0 included)
            raw = broadcast(bytes(ident) if rank == 0 else None)
            ident = (C.c_uint8 * 128).from_buffer_copy(raw)
        h = C.c_void_p()
        _lib.check(L, L.irs_hip_comm_init_rank(device_index, ident, world, rank, C.byref(h)),
                   "irs_hip_comm_init_rank")
        self.handle = h
        self.ranks_seen = None   # set by agreed_communicator's self-test all-gather

    def library(self) -> str:
        """Which RCCL the communicator runs on ("mapped:<path>": the copy the process — torch —
        had loaded already)."""
        buf = C.create_string_buffer(600)
        _lib.check(self.L, self.L.irs_hip_comm_library(buf, len(buf)), "irs_hip_comm_library")
        return buf.value.decode()

    def all_gather(self, d_send: int, d_recv: int, bytes_per_rank: int, stream=None):
        _lib.check(self.L, self.L.irs_hip_topk_allgather(self.handle, d_send, d_recv,
                                                         bytes_per_rank, stream),
                   "irs_hip_topk_allgather")

    def close(self):
        if self.handle:
            self.L.irs_hip_comm_destroy(self.handle)
            self.handle = None


def agreed_communicator(L, device_index: int, rank: int, world: int, device, log=None):
    """Communicator on EVERY rank or on none: a rank that cannot bind RCCL (or whose
    ncclCommInitRank fails) must not leave the others waiting in a collective it never joins.
    Each step is followed by an all-reduce(MIN) of the ranks' success flags over
    torch.distributed; on any failure every rank drops its communicator and the caller falls
    back to torch.distributed's all-gather.  Returns the communicator or None."""
    def all_ok(ok: bool) -> bool:
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    why = None
    ident = (C.c_uint8 * 128)()
    try:   # capability probe on every rank (rank 0's id is the one that counts)
        _lib.check(L, L.irs_hip_comm_unique_id(ident), "irs_hip_comm_unique_id")
        ok = True
    except Exception as e:  # noqa: BLE001
        ok, why = False, e
    if not all_ok(ok):
        if log:
            log("irs_hip_comm unavailable on some rank (%s): all-gather through torch.distributed" % why)
        return None
    box = [bytes(ident) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    comm = None
    try:
        comm = Communicator(L, device_index, rank, world, broadcast=lambda _b: box[0])
        ok = True
    except Exception as e:  # noqa: BLE001
        ok, why = False, e
    if not all_ok(ok):
        if comm is not None:
            comm.close()
        if log:
            log("irs_hip_comm_init_rank failed on some rank (%s): all-gather through torch.distributed" % why)
        return None
    # one small all-gather with known contents before anything depends on it
    try:
        n = 256
        send = torch.full((n,), rank + 1, dtype=torch.int32, device=device)
        recv = torch.zeros((world * n,), dtype=torch.int32, device=device)
        stream = None
        if device.type == "cuda":
            stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        comm.all_gather(send.data_ptr(), recv.data_ptr(), 4 * n, stream)
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        want = torch.arange(1, world + 1, dtype=torch.int32, device=device).repeat_interleave(n)
        ok = bool(torch.equal(recv, want))
        why = "self-test all-gather returned wrong data"
        # what the self-test saw: how many distinct ranks' blocks arrived (bench.py reports it)
        comm.ranks_seen = int(torch.unique(recv).numel())
    except Exception as e:  # noqa: BLE001
        ok, why = False, e
    if not all_ok(ok):
        comm.close()
        if log:
            log("irs_hip_topk_allgather self-test failed on some rank (%s): torch.distributed" % why)
        return None
    return comm


class _CommWork:
    """What `gather(async_op=True)` returns on the C-ABI path: wait() makes the current stream
    wait for the collective (like the work handle of torch.distributed)."""

    def __init__(self, done):
        self.done = done

    def wait(self):
        if self.done is not None:
            torch.cuda.current_stream().wait_event(self.done)


class TopkExchange:
    """The one exchange step of the path, with every buffer allocated once.

    Each rank owns `per` segment slots.  Its send buffer is ONE int64 tensor: `per` hit
    tables [nq][k] (an int64 carries one irs_hip_hit) followed by `per` count tables [nq]
    (int32, two per int64) — `slot(i)` hands out the device pointers of slot i so that
    irs_hip_batch_results_to_device writes straight into it.  `run()` = one
    all_gather_into_tensor (RCCL over xGMI with backend "nccl") + irs_hip_merge_topk.
    """

    def __init__(self, L, device_index: int, n_segments: int, rank: int, world: int, nq: int,
                 k: int, tensor_device, comm: "Communicator | None" = None):
        self.L, self.device_index, self.nq, self.k = L, device_index, nq, k
        self.world, self.rank = world, rank
        # comm: the collective goes through irs_hip_topk_allgather (RCCL called from the
        # library, on its own stream) instead of torch.distributed
        self.comm = comm
        self.on_gpu = torch.device(tensor_device).type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=tensor_device) if (comm and self.on_gpu) else None
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
        if self.world <= 1:
            return None
        if self.comm is None:
            return dist.all_gather_into_tensor(self.recv, self.send, async_op=async_op)
        nbytes = self.send.numel() * 8
        if not self.on_gpu:   # emulator tier: host buffers, no streams
            self.comm.all_gather(self.send.data_ptr(), self.recv.data_ptr(), nbytes)
            return _CommWork(None) if async_op else None
        cur = torch.cuda.current_stream()
        self.comm_stream.wait_stream(cur)            # the send buffer is complete
        self.comm.all_gather(self.send.data_ptr(), self.recv.data_ptr(), nbytes,
                             C.c_void_p(self.comm_stream.cuda_stream))
        done = torch.cuda.Event()
        done.record(self.comm_stream)
        if async_op:
            return _CommWork(done)
        cur.wait_event(done)
        return None

    def merge(self, stream=None):
        _lib.check(self.L, self.L.irs_hip_merge_topk(
            self.device_index, self._lists, self._counts, self._seg_ids.ctypes.data,
            self.n_lists, self.nq, self.k, self.out_h.data_ptr(), self.out_s.data_ptr(),
            self.out_c.data_ptr(), stream), "irs_hip_merge_topk")
        return self.out_h, self.out_s, self.out_c

    def run(self, stream=None):
        self.gather()
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
