"""ctypes binding of the C ABI in include/irs_hip.h.

The product library is iresearch_amd/csrc/libirs_hip.so (hipcc, gfx950).  It is
the only library this module ever loads by itself, and loading it — or opening a
segment on a machine without a gfx950 GPU — fails loudly: there is no CPU
fallback.  (`bind()` is exposed so that the CPU-only test tier can attach the
same prototypes to the fiber-emulator build under tests/sim.)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _build

OK, EINVAL, ECORRUPT, ENOMEM, EHIP, EOVERFLOW, EUNSUPPORTED, EPEER = 0, -1, -2, -3, -4, -5, -6, -7
LAYOUT_SCALAR, LAYOUT_SIMD4 = 0, 1
OP_OR, OP_AND, OP_MINMATCH, OP_PHRASE = 0, 1, 2, 3
SCORE_BM25, SCORE_BM15, SCORE_BM1, SCORE_TFIDF, SCORE_TFIDF_NORM = 0, 1, 2, 3, 4
NO_TERM = 0xFFFFFFFF
PATH_AUTO, PATH_ITEMS, PATH_JOINED = 0, 1, 2
WAND_NONE, WAND_DIV_NORM, WAND_MAX_FREQ, WAND_MIN_NORM = 0, 1, 2, 3   # Scorer::WandType
MAX_TERMS, MAX_K, MAX_PHRASE_TERMS = 16, 4096, 8
K_PLAN, K_PILOT, K_SCORE, K_SELECT, K_COUNT = 0, 1, 2, 3, 4
KERNEL_NAMES = ("k_plan", "k_pilot", "k_score", "k_select")
KERNEL_NAMES_JOINED = ("k_join", "k_join_pilot", "k_join_score", "k_select")

TERM_META = np.dtype(
    [("docs_count", "<u4"), ("freq", "<u4"), ("doc_start", "<u8"), ("pos_start", "<u8"),
     ("pos_end", "<u8"), ("pay_start", "<u8"), ("e_skip_start", "<u8")], align=True)
TERM_SCORER = np.dtype(
    [("term", "<u4"), ("kind", "<i4"), ("c0", "<f4"), ("norm_const", "<f4"),
     ("norm_length", "<f4"), ("phrase_offset", "<u4")], align=True)
QUERY = np.dtype([("op", "<i4"), ("n_terms", "<u4"), ("first_term", "<u4"), ("k", "<u4"),
                  ("min_match", "<u4"), ("merge", "<u4")], align=True)
HIT = np.dtype([("score", "<f4"), ("doc", "<u4")], align=True)
assert TERM_META.itemsize == 48 and TERM_SCORER.itemsize == 24
assert QUERY.itemsize == 24 and HIT.itemsize == 8


class SegmentDesc(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("layout", C.c_int32), ("doc_file", C.c_void_p),
        ("doc_file_len", C.c_uint64), ("num_docs", C.c_uint32), ("has_freq", C.c_uint32),
        ("norms", C.c_void_p), ("norm_width", C.c_uint32), ("norm_min_doc", C.c_uint32),
        ("norm_count", C.c_uint64), ("terms", C.c_void_p), ("num_terms", C.c_uint32),
        ("wand_count", C.c_uint32), ("pos_file", C.c_void_p), ("pos_file_len", C.c_uint64),
        ("pos_features", C.c_uint32), ("norm_kind", C.c_uint32), ("wand_type", C.c_uint32),
        ("doc_mask", C.c_void_p), ("doc_mask_count", C.c_uint64),
    ]


class IrsHipError(RuntimeError):
    def __init__(self, status: int, what: str, message: str):
        super().__init__("%s: %s (%d)" % (what, message, status))
        self.status = status


SYMBOLS = (
    "irs_hip_abi_version", "irs_hip_strerror", "irs_hip_device_arch", "irs_hip_segment_open",
    "irs_hip_segment_close", "irs_hip_segment_device_bytes", "irs_hip_segment_live_docs",
    "irs_hip_decode_term",
    "irs_hip_decode_positions",
    "irs_hip_term_directory", "irs_hip_bit_union", "irs_hip_bit_union_counts", "irs_hip_batch_create",
    "irs_hip_batch_create_multi", "irs_hip_batch_run",
    "irs_hip_batch_results", "irs_hip_batch_device_results",
    "irs_hip_batch_results_to_host", "irs_hip_batch_host_results",
    "irs_hip_batch_results_to_device", "irs_hip_batch_destroy",
    "irs_hip_query_batch", "irs_hip_batch_configure", "irs_hip_batch_profile",
    "irs_hip_batch_timings", "irs_hip_batch_work", "irs_hip_batch_reruns", "irs_hip_merge_topk",
    "irs_hip_batch_plan", "irs_hip_batch_set_wand", "irs_hip_batch_set_min_scores",
    "irs_hip_batch_set_path", "irs_hip_batch_path", "irs_hip_batch_set_shared_threshold",
    "irs_hip_batch_set_async", "irs_hip_batch_set_paired_tiles", "irs_hip_batch_paired_tiles",
    "irs_hip_batch_set_comm",
    "irs_hip_term_blockmax",
    "irs_hip_segment_wand_source",
    "irs_hip_batch_touched",
    "irs_hip_comm_unique_id", "irs_hip_comm_init_rank", "irs_hip_comm_destroy",
    "irs_hip_comm_library",
    "irs_hip_topk_allgather", "irs_hip_device_alloc", "irs_hip_device_free",
    "irs_hip_device_upload", "irs_hip_device_download", "irs_hip_device_sync",
    "irs_hip_device_trim",
)


def bind(L: C.CDLL) -> C.CDLL:
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
    P = C.POINTER
    L.irs_hip_abi_version.argtypes, L.irs_hip_abi_version.restype = [], u32
    L.irs_hip_strerror.argtypes, L.irs_hip_strerror.restype = [C.c_int], C.c_char_p
    L.irs_hip_device_arch.argtypes, L.irs_hip_device_arch.restype = [i32, C.c_char_p, C.c_size_t], C.c_int
    L.irs_hip_segment_open.argtypes = [P(SegmentDesc), P(vp)]
    L.irs_hip_segment_open.restype = C.c_int
    L.irs_hip_segment_close.argtypes, L.irs_hip_segment_close.restype = [vp], None
    L.irs_hip_segment_device_bytes.argtypes = [vp]
    L.irs_hip_segment_device_bytes.restype = u64
    L.irs_hip_segment_live_docs.argtypes = [vp]
    L.irs_hip_segment_live_docs.restype = u64
    L.irs_hip_decode_term.argtypes = [vp, u32, vp, vp, u32, P(u32)]
    L.irs_hip_decode_term.restype = C.c_int
    L.irs_hip_decode_positions.argtypes = [vp, u32, vp, u64, P(u64)]
    L.irs_hip_decode_positions.restype = C.c_int
    L.irs_hip_bit_union.argtypes = [vp, vp, u32, vp, u64, P(u64)]
    L.irs_hip_bit_union.restype = C.c_int
    L.irs_hip_bit_union_counts.argtypes = [vp, vp, vp, u32, vp]
    L.irs_hip_bit_union_counts.restype = C.c_int
    L.irs_hip_term_directory.argtypes = [vp, u32, vp, vp, u32, P(u32)]
    L.irs_hip_term_directory.restype = C.c_int
    L.irs_hip_batch_create.argtypes = [vp, vp, u32, vp, u32, P(vp)]
    L.irs_hip_batch_create.restype = C.c_int
    L.irs_hip_batch_create_multi.argtypes = [vp, u32, vp, u32, vp, u32, P(vp)]
    L.irs_hip_batch_create_multi.restype = C.c_int
    L.irs_hip_batch_run.argtypes, L.irs_hip_batch_run.restype = [vp, vp], C.c_int
    L.irs_hip_batch_results_to_host.argtypes = [vp, vp]
    L.irs_hip_batch_results_to_host.restype = C.c_int
    L.irs_hip_batch_host_results.argtypes = [vp, P(vp), P(u32), P(vp), P(vp)]
    L.irs_hip_batch_host_results.restype = C.c_int
    L.irs_hip_batch_results.argtypes = [vp, vp, u32, vp, vp]
    L.irs_hip_batch_results.restype = C.c_int
    L.irs_hip_batch_device_results.argtypes = [vp, P(vp), P(vp), P(u32)]
    L.irs_hip_batch_device_results.restype = C.c_int
    L.irs_hip_batch_results_to_device.argtypes = [vp, vp, vp, vp]
    L.irs_hip_batch_results_to_device.restype = C.c_int
    L.irs_hip_batch_destroy.argtypes, L.irs_hip_batch_destroy.restype = [vp], None
    L.irs_hip_query_batch.argtypes = [vp, vp, u32, vp, u32, vp, u32, vp, vp]
    L.irs_hip_query_batch.restype = C.c_int
    L.irs_hip_batch_configure.argtypes = [vp, u32, u32, u32]
    L.irs_hip_batch_configure.restype = C.c_int
    L.irs_hip_batch_profile.argtypes, L.irs_hip_batch_profile.restype = [vp, C.c_int], C.c_int
    L.irs_hip_batch_timings.argtypes = [vp, P(C.c_float)]
    L.irs_hip_batch_timings.restype = C.c_int
    L.irs_hip_batch_reruns.argtypes, L.irs_hip_batch_reruns.restype = [vp, P(u32)], C.c_int
    L.irs_hip_batch_work.argtypes = [vp, P(u64), P(u64)]
    L.irs_hip_batch_work.restype = C.c_int
    L.irs_hip_merge_topk.argtypes = [i32, vp, vp, vp, u32, u32, u32, vp, vp, vp, vp]
    L.irs_hip_merge_topk.restype = C.c_int
    L.irs_hip_batch_set_wand.argtypes, L.irs_hip_batch_set_wand.restype = [vp, C.c_int], C.c_int
    L.irs_hip_batch_plan.argtypes, L.irs_hip_batch_plan.restype = [vp, vp], C.c_int
    L.irs_hip_batch_set_path.argtypes, L.irs_hip_batch_set_path.restype = [vp, C.c_int], C.c_int
    L.irs_hip_batch_path.argtypes, L.irs_hip_batch_path.restype = [vp, P(C.c_int)], C.c_int
    L.irs_hip_batch_set_async.argtypes, L.irs_hip_batch_set_async.restype = [vp, C.c_int], C.c_int
    L.irs_hip_batch_set_paired_tiles.argtypes = [vp, C.c_int]
    L.irs_hip_batch_set_paired_tiles.restype = C.c_int
    L.irs_hip_batch_paired_tiles.argtypes, L.irs_hip_batch_paired_tiles.restype = [vp, P(C.c_int)], C.c_int
    L.irs_hip_batch_set_shared_threshold.argtypes = [vp, C.c_int]
    L.irs_hip_batch_set_shared_threshold.restype = C.c_int
    L.irs_hip_batch_set_comm.argtypes, L.irs_hip_batch_set_comm.restype = [vp, vp], C.c_int
    L.irs_hip_segment_wand_source.argtypes = [vp, P(u64), P(u64)]
    L.irs_hip_segment_wand_source.restype = C.c_int
    L.irs_hip_batch_set_min_scores.argtypes = [vp, vp]
    L.irs_hip_batch_set_min_scores.restype = C.c_int
    L.irs_hip_term_blockmax.argtypes = [vp, u32, vp, vp, u32, P(u32)]
    L.irs_hip_term_blockmax.restype = C.c_int
    L.irs_hip_batch_touched.argtypes, L.irs_hip_batch_touched.restype = [vp, P(u64), P(u64)], C.c_int
    L.irs_hip_comm_unique_id.argtypes, L.irs_hip_comm_unique_id.restype = [vp], C.c_int
    L.irs_hip_comm_init_rank.argtypes = [i32, vp, i32, i32, P(vp)]
    L.irs_hip_comm_init_rank.restype = C.c_int
    L.irs_hip_comm_destroy.argtypes, L.irs_hip_comm_destroy.restype = [vp], None
    L.irs_hip_comm_library.argtypes, L.irs_hip_comm_library.restype = [C.c_char_p, C.c_size_t], C.c_int
    L.irs_hip_topk_allgather.argtypes = [vp, vp, vp, u64, vp]
    L.irs_hip_topk_allgather.restype = C.c_int
    L.irs_hip_device_trim.argtypes, L.irs_hip_device_trim.restype = [i32], C.c_int
    return L


_lib = None


def lib() -> C.CDLL:
    """The product library. Raises if it cannot be built/loaded."""
    global _lib
    if _lib is None:
        try:
            # One HIP runtime per process: torch bundles its own libamdhip64.so.7;
            # loading it first makes libirs_hip.so (same SONAME) share it, so that
            # torch.distributed/RCCL tensors and our kernels see the same device context.
            import torch  # noqa: F401
        except ImportError:  # the C ABI itself does not need torch
            pass
        _lib = bind(C.CDLL(str(_build.build_hip())))
    return _lib


def check(L: C.CDLL, status: int, what: str):
    if status != OK:
        raise IrsHipError(status, what, L.irs_hip_strerror(status).decode())
