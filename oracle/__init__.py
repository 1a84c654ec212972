"""oracle — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
only; nothing under iresearch_amd/ imports this package (tests/test_layout.py
enforces it).  See oracle/oracle.h for what is pinned against what.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent

LAYOUT_SCALAR, LAYOUT_SIMD4 = 0, 1
SCORER_BM25, SCORER_TFIDF = 0, 1
OP_OR, OP_AND, OP_MINMATCH = 0, 1, 2  # MINMATCH: op | (min_match << 8)

TERM_META = np.dtype(
    [("docs_count", "<u4"), ("freq", "<u4"), ("doc_start", "<u8"), ("pos_start", "<u8"),
     ("pos_end", "<u8"), ("pay_start", "<u8"), ("e_skip_start", "<u8")],
    align=True,
)
HIT = np.dtype([("score", "<f4"), ("doc", "<u4"), ("segment", "<u4")])


class Segment(C.Structure):
    _fields_ = [("doc_file", C.c_void_p), ("doc_file_len", C.c_uint64), ("layout", C.c_int32),
                ("num_docs", C.c_uint32), ("norms", C.c_void_p), ("norm_width", C.c_uint32),
                ("wand_count", C.c_uint32), ("pos_file", C.c_void_p), ("pos_file_len", C.c_uint64),
                ("pos_one_based", C.c_int32), ("doc_mask", C.c_void_p),
                ("doc_mask_count", C.c_uint64)]


class Scorer(C.Structure):
    _fields_ = [("kind", C.c_int32), ("k", C.c_float), ("b", C.c_float),
                ("with_norms", C.c_int32)]


class BM25Stats(C.Structure):
    _fields_ = [("idf", C.c_float), ("norm_const", C.c_float), ("norm_length", C.c_float),
                ("norm_cache", C.c_float * 256)]


NORM_LEGACY_F32 = 0x104  # orc_segment.norm_width: legacy `Norm` floats (oracle.h)

_lib = None
_ref = None


def build(force: bool = False):
    deps = [HERE / n for n in ("postings_oracle.c", "search_oracle.cpp", "dict_oracle.cpp",
                               "oracle.h", "oracle_internal.h")]
    lib = HERE / "liboracle.so"
    if force or not lib.exists() or any(d.stat().st_mtime > lib.stat().st_mtime for d in deps):
        subprocess.run(["make", "-C", str(HERE), "liboracle.so"], check=True,
                       capture_output=True)
    return lib


def build_native():
    """The same sources built the way SURVEY.md §8d states the CPU baseline: -O3
    -march=native (bench.py's `cpu_baseline` leg only; compiled where it runs, since
    "native" of the build container need not exist on the GPU box).  None if that fails."""
    lib = HERE / "liboracle_native.so"
    try:
        subprocess.run(["make", "-C", str(HERE), "-B", "liboracle_native.so"], check=True,
                       capture_output=True)
    except (OSError, subprocess.CalledProcessError):
        return None
    return lib if lib.exists() else None


def use_native() -> bool:
    """Switches every call of this module to the -O3 -march=native build (bench only)."""
    global _lib
    so = build_native()
    if so is None:
        return False
    _lib = None
    lib(so)
    return True


def lib(path=None):
    global _lib
    if _lib is None:
        L = C.CDLL(str(path or build()))
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
        for n in ("orc_pack_scalar", "orc_unpack_scalar", "orc_pack_simd4", "orc_unpack_simd4"):
            getattr(L, n).argtypes = [vp, u32, vp]
            getattr(L, n).restype = None
        L.orc_at_scalar.argtypes = [vp, u32, u32]
        L.orc_at_scalar.restype = u32
        L.orc_read_block.argtypes = [vp, vp, C.c_int, vp]
        L.orc_read_block.restype = C.c_int64
        L.orc_decode_term.argtypes = [vp, u64, C.c_int, vp, vp, vp, u64]
        L.orc_decode_term.restype = C.c_int64
        L.orc_decode_term_wand.argtypes = [vp, u64, C.c_int, C.c_int, u32, vp, vp, vp, u64]
        L.orc_decode_term_wand.restype = C.c_int64
        L.orc_bit_union_wand.argtypes = [vp, u64, C.c_int, C.c_int, u32, vp, u32, vp, u64]
        L.orc_bit_union_wand.restype = C.c_int64
        L.orc_read_skip0_wand.argtypes = [vp, u64, u32, vp, vp, vp, u64, C.POINTER(u32), vp, vp]
        L.orc_read_skip0_wand.restype = C.c_int64
        L.orc_bit_union.argtypes = [vp, u64, C.c_int, C.c_int, vp, u32, vp, u64]
        L.orc_bit_union.restype = C.c_int64
        L.orc_read_skip0.argtypes = [vp, u64, vp, vp, vp, u64, C.POINTER(u32)]
        L.orc_read_skip0.restype = C.c_int64
        L.orc_check_doc_header.argtypes = [vp, u64, C.POINTER(i32)]
        L.orc_check_doc_header.restype = C.c_int64
        L.orc_bm25_collect.argtypes = [C.c_float, C.c_float, u64, u64, u64, C.POINTER(BM25Stats)]
        L.orc_bm25_collect.restype = None
        L.orc_tfidf_idf.argtypes = [u64, u64]
        L.orc_tfidf_idf.restype = C.c_float
        L.orc_search.argtypes = [vp, u32, vp, u32, i32, C.POINTER(Scorer), vp, vp, vp, u32, vp,
                                 C.POINTER(u64)]
        L.orc_search.restype = C.c_int64
        L.orc_search_batch.argtypes = [vp, u32, vp, u32, u32, i32, C.POINTER(Scorer), vp, vp,
                                       u32, u32, vp, vp, vp]
        L.orc_search_batch.restype = C.c_int64
        L.orc_score_all.argtypes = [C.POINTER(Segment), vp, u32, i32, C.POINTER(Scorer), vp, u64,
                                    vp, u64, vp, vp]
        L.orc_score_all.restype = C.c_int64
        L.orc_read_skip0_pos.argtypes = [vp, u64, u32, C.c_int, vp, vp, vp, u64, C.POINTER(u32),
                                         vp, vp, vp, vp]
        L.orc_read_skip0_pos.restype = C.c_int64
        L.orc_decode_positions_v.argtypes = [vp, u64, vp, u64, C.c_int, u32, C.c_int, vp, u32, vp,
                                             u64]
        L.orc_decode_positions_v.restype = C.c_int64
        L.orc_check_pos_header.argtypes = [vp, u64, C.POINTER(i32)]
        L.orc_check_pos_header.restype = C.c_int64
        L.orc_search_phrase.argtypes = [vp, u32, vp, u32, vp, C.POINTER(Scorer), C.c_float, vp, vp,
                                        u32, vp, C.POINTER(u64)]
        L.orc_search_phrase.restype = C.c_int64
        L.orc_score_all_phrase.argtypes = [C.POINTER(Segment), vp, u32, vp, C.POINTER(Scorer),
                                           C.c_float, u64, vp, u64, vp, vp]
        L.orc_score_all_phrase.restype = C.c_int64
        L.orc_encode_term_meta.argtypes = [vp, vp, C.c_int, C.c_int, vp, u64]
        L.orc_encode_term_meta.restype = C.c_int64
        L.orc_decode_term_meta.argtypes = [vp, u64, C.c_int, C.c_int, C.c_int, vp]
        L.orc_decode_term_meta.restype = C.c_int64
        L.orc_read_document_mask.argtypes = [vp, u64, vp, u64]
        L.orc_read_document_mask.restype = C.c_int64
        L.orc_walk_term_dictionary.argtypes = [vp, u64, u64, C.c_int, C.c_int, C.c_int,
                                               C.POINTER(u32), C.POINTER(u64), vp, vp, vp]
        L.orc_walk_term_dictionary.restype = C.c_int
        L.orc_read_fixed_column.argtypes = [vp, u64, vp, u64, u32, C.POINTER(u32), C.POINTER(u32),
                                            C.POINTER(u32), vp, u32, C.POINTER(u32), vp, u64]
        L.orc_read_fixed_column.restype = C.c_int
        _lib = L
    return _lib


def ref():
    """The reference's own compiled codec (oracle/_ref), or None when absent."""
    global _ref
    if _ref is None:
        so = HERE / "_ref" / "libref_codec.so"
        if not so.exists() and Path("/root/reference/core/utils").is_dir():
            subprocess.run(["make", "-C", str(HERE), "ref"], check=True, capture_output=True)
        if not so.exists():
            return None
        R = C.CDLL(str(so))
        for n in ("ref_pack_scalar", "ref_unpack_scalar", "ref_pack_simd4", "ref_unpack_simd4"):
            getattr(R, n).argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
            getattr(R, n).restype = None
        R.ref_at_scalar.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        R.ref_at_scalar.restype = C.c_uint32
        _ref = R
    return _ref


# ----------------------------------------------------------------- helpers --

def pack(values, bits: int, layout: int) -> np.ndarray:
    v = np.ascontiguousarray(values, np.uint32)
    assert v.size == 128
    out = np.zeros(4 * bits, np.uint32)
    f = lib().orc_pack_simd4 if layout == LAYOUT_SIMD4 else lib().orc_pack_scalar
    f(v.ctypes.data, bits, out.ctypes.data)
    return out


def unpack(words, bits: int, layout: int) -> np.ndarray:
    w = np.ascontiguousarray(words, np.uint32)
    out = np.zeros(128, np.uint32)
    f = lib().orc_unpack_simd4 if layout == LAYOUT_SIMD4 else lib().orc_unpack_scalar
    f(w.ctypes.data, bits, out.ctypes.data)
    return out


def decode_term(doc_file: np.ndarray, meta, layout: int, want_freq: bool = True,
                field_has_freq: bool = True, wand_count: int = 0):
    m = np.zeros(1, TERM_META)
    for k in TERM_META.names:
        m[0][k] = meta[k]
    n = int(m[0]["docs_count"])
    docs = np.zeros(n, np.uint32)
    freqs = np.zeros(n, np.uint32) if want_freq else None
    got = lib().orc_decode_term_wand(doc_file.ctypes.data, doc_file.size, layout,
                                     int(field_has_freq), wand_count, m.ctypes.data,
                                     docs.ctypes.data, freqs.ctypes.data if want_freq else None, n)
    if got != n:
        raise ValueError("orc_decode_term: %d != %d" % (got, n))
    return docs, freqs


def decode_positions(doc_file: np.ndarray, pos_file: np.ndarray, meta, layout: int,
                     wand_count: int = 0, stride: int = 1, one_based: bool = False) -> np.ndarray:
    """All positions of one term, doc after doc (Σ freq values); stride > 1 drains every
    stride-th doc only (the rest goes through position::skip and reads as 0)."""
    m = np.zeros(1, TERM_META)
    for k in TERM_META.names:
        m[0][k] = meta[k]
    n = int(m[0]["freq"])
    out = np.zeros(n, np.uint32)
    got = lib().orc_decode_positions_v(doc_file.ctypes.data, doc_file.size, pos_file.ctypes.data,
                                       pos_file.size, layout, wand_count, int(one_based),
                                       m.ctypes.data, stride, out.ctypes.data, n)
    if got != n:
        raise ValueError("orc_decode_positions: %d != %d" % (got, n))
    return out


def read_skip0_pos(doc_file: np.ndarray, meta, wand_count: int = 0):
    """Level-0 skip entries of a field with POS: (last docs, doc ptrs, pend_pos, pos ptrs)."""
    m = np.zeros(1, TERM_META)
    for k in TERM_META.names:
        m[0][k] = meta[k]
    cap = int(m[0]["docs_count"]) // 128 + 1
    last = np.zeros(cap, np.uint32)
    ptrs = np.zeros(cap, np.uint64)
    pend = np.zeros(cap, np.uint32)
    pptr = np.zeros(cap, np.uint64)
    lv = C.c_uint32()
    n = lib().orc_read_skip0_pos(doc_file.ctypes.data, doc_file.size, wand_count, 1,
                                 m.ctypes.data, last.ctypes.data, ptrs.ctypes.data, cap,
                                 C.byref(lv), None, None, pend.ctypes.data, pptr.ctypes.data)
    if n < 0:
        raise ValueError("orc_read_skip0_pos failed: %d" % n)
    return last[:n], ptrs[:n], pend[:n], pptr[:n]


def bit_union(doc_file: np.ndarray, metas, layout: int, has_freq: bool, n_words: int,
              initial: np.ndarray | None = None, wand_count: int = 0):
    if isinstance(metas, np.ndarray) and metas.dtype == TERM_META:
        m = np.ascontiguousarray(metas)
    else:
        m = np.zeros(len(metas), TERM_META)
        for i, meta in enumerate(metas):
            for k in TERM_META.names:
                m[i][k] = meta[k]
    bits = np.zeros(n_words, np.uint64) if initial is None else initial.copy()
    n = lib().orc_bit_union_wand(doc_file.ctypes.data, doc_file.size, layout, int(has_freq),
                                 wand_count, m.ctypes.data, len(metas), bits.ctypes.data, n_words)
    if n < 0:
        raise ValueError("orc_bit_union failed: %d" % n)
    return bits, int(n)


def read_skip0(doc_file: np.ndarray, meta, wand_count: int = 0, with_wand: bool = False,
               has_pos: bool = False):
    """Level-0 skip entries: (last docs, next block pointers, #levels) and, with_wand, the
    (max freq, norm) payload of scorer 0 in every entry.  has_pos: the field stores positions
    (the entries then also carry pend_pos and the `.pos` pointer)."""
    m = np.zeros(1, TERM_META)
    for k in TERM_META.names:
        m[0][k] = meta[k]
    cap = int(m[0]["docs_count"]) // 128 + 1
    last = np.zeros(cap, np.uint32)
    ptrs = np.zeros(cap, np.uint64)
    mf = np.zeros(cap, np.uint32)
    nm = np.zeros(cap, np.uint32)
    lv = C.c_uint32()
    n = lib().orc_read_skip0_pos(doc_file.ctypes.data, doc_file.size, wand_count,
                                 1 if has_pos else 0, m.ctypes.data, last.ctypes.data,
                                 ptrs.ctypes.data, cap, C.byref(lv), mf.ctypes.data,
                                 nm.ctypes.data, None, None)
    if n < 0:
        raise ValueError("orc_read_skip0 failed: %d" % n)
    if with_wand:
        return last[:n], ptrs[:n], lv.value, mf[:n], nm[:n]
    return last[:n], ptrs[:n], lv.value


def bm25_stats(k: float, b: float, docs_with_field: int, docs_with_term: int,
               total_term_freq: int) -> BM25Stats:
    st = BM25Stats()
    lib().orc_bm25_collect(k, b, docs_with_field, docs_with_term, total_term_freq, C.byref(st))
    return st


class SegmentView:
    """Keeps the numpy buffers of one segment alive next to the C struct."""

    def __init__(self, doc_file, norms, layout, num_docs, docs_with_field, total_term_freq,
                 norm_width=1, wand_count=0, pos_file=None, pos_one_based=False, doc_mask=None):
        self.doc_file = np.ascontiguousarray(doc_file, np.uint8)
        self.norms = None if norms is None else np.ascontiguousarray(norms, np.uint8)
        self.layout, self.num_docs, self.norm_width = layout, num_docs, norm_width
        self.docs_with_field, self.total_term_freq = docs_with_field, total_term_freq
        self.wand_count = wand_count
        self.pos_file = None if pos_file is None else np.ascontiguousarray(pos_file, np.uint8)
        self.pos_one_based = bool(pos_one_based)
        # the segment's DocumentMask (deleted doc ids): every search masks its iterator with it
        self.doc_mask = None if doc_mask is None else np.ascontiguousarray(doc_mask, np.uint32)

    def struct(self) -> Segment:
        return Segment(self.doc_file.ctypes.data, self.doc_file.size, self.layout, self.num_docs,
                       None if self.norms is None else self.norms.ctypes.data, self.norm_width,
                       self.wand_count,
                       None if self.pos_file is None else self.pos_file.ctypes.data,
                       0 if self.pos_file is None else self.pos_file.size,
                       int(self.pos_one_based),
                       None if self.doc_mask is None or not self.doc_mask.size else self.doc_mask.ctypes.data,
                       0 if self.doc_mask is None else self.doc_mask.size)


def _metas_array(metas) -> np.ndarray:
    src = np.asarray(metas)
    out = np.zeros(src.shape, TERM_META)
    for k in TERM_META.names:
        out[k] = src[k]
    return np.ascontiguousarray(out)


def search(segments, metas, op: int, scorer: Scorer, k: int, boosts=None):
    """metas: TERM_META array [nsegs][n_terms]. Returns (hits HIT[], total)."""
    metas = _metas_array(metas)
    nsegs, n_terms = metas.shape
    segs = (Segment * nsegs)(*[s.struct() for s in segments])
    dwf = np.array([s.docs_with_field for s in segments], np.uint64)
    ttf = np.array([s.total_term_freq for s in segments], np.uint64)
    out = np.zeros(max(k, 1), HIT)
    total = C.c_uint64()
    b = None if boosts is None else np.ascontiguousarray(boosts, np.float32)
    n = lib().orc_search(segs, nsegs, metas.ctypes.data, n_terms, op, C.byref(scorer),
                         None if b is None else b.ctypes.data, dwf.ctypes.data, ttf.ctypes.data,
                         k, out.ctypes.data, C.byref(total))
    if n < 0:
        raise ValueError("orc_search failed")
    return out[:n], total.value


def search_batch(segments, metas, op: int, scorer: Scorer, k: int, threads: int = 1):
    """metas: TERM_META array [n_queries][nsegs][n_terms]."""
    metas = _metas_array(metas)
    nq, nsegs, n_terms = metas.shape
    segs = (Segment * nsegs)(*[s.struct() for s in segments])
    dwf = np.array([s.docs_with_field for s in segments], np.uint64)
    ttf = np.array([s.total_term_freq for s in segments], np.uint64)
    out = np.zeros((nq, max(k, 1)), HIT)
    counts = np.zeros(nq, np.uint32)
    totals = np.zeros(nq, np.uint64)
    n = lib().orc_search_batch(segs, nsegs, metas.ctypes.data, nq, n_terms, op,
                               C.byref(scorer), dwf.ctypes.data, ttf.ctypes.data, k, threads,
                               out.ctypes.data, counts.ctypes.data, totals.ctypes.data)
    if n < 0:
        raise ValueError("orc_search_batch failed")
    return out, counts, totals


def score_all(segment: SegmentView, metas, op: int, scorer: Scorer, docs_with_field: int,
              docs_with_term, total_term_freq: int, boosts=None):
    metas = _metas_array(metas)
    n_terms = metas.shape[0]
    seg = segment.struct()
    scores = np.zeros(segment.num_docs + 1, np.float32)
    matched = np.zeros(segment.num_docs + 1, np.uint8)
    dwt = np.ascontiguousarray(docs_with_term, np.uint64)
    b = None if boosts is None else np.ascontiguousarray(boosts, np.float32)
    n = lib().orc_score_all(C.byref(seg), metas.ctypes.data, n_terms, op, C.byref(scorer),
                            None if b is None else b.ctypes.data, docs_with_field,
                            dwt.ctypes.data, total_term_freq, scores.ctypes.data,
                            matched.ctypes.data)
    if n < 0:
        raise ValueError("orc_score_all failed")
    return scores, matched



def search_phrase(segments, metas, offsets, scorer: Scorer, k: int, boost: float = 1.0):
    """by_phrase with fixed offsets; metas: TERM_META [nsegs][n_terms]. -> (hits HIT[], total)."""
    metas = _metas_array(metas)
    nsegs, n_terms = metas.shape
    segs = (Segment * nsegs)(*[s.struct() for s in segments])
    dwf = np.array([s.docs_with_field for s in segments], np.uint64)
    ttf = np.array([s.total_term_freq for s in segments], np.uint64)
    offs = np.ascontiguousarray(offsets, np.uint32)
    assert offs.size == n_terms
    out = np.zeros(max(k, 1), HIT)
    total = C.c_uint64()
    n = lib().orc_search_phrase(segs, nsegs, metas.ctypes.data, n_terms, offs.ctypes.data,
                                C.byref(scorer), boost, dwf.ctypes.data, ttf.ctypes.data, k,
                                out.ctypes.data, C.byref(total))
    if n < 0:
        raise ValueError("orc_search_phrase failed")
    return out[:n], total.value


def score_all_phrase(segment: SegmentView, metas, offsets, scorer: Scorer, docs_with_field: int,
                     docs_with_term, total_term_freq: int, boost: float = 1.0):
    """-> (scores f32[num_docs + 1], phrase_freq u32[num_docs + 1])."""
    metas = _metas_array(metas)
    n_terms = metas.shape[0]
    seg = segment.struct()
    scores = np.zeros(segment.num_docs + 1, np.float32)
    pf = np.zeros(segment.num_docs + 1, np.uint32)
    dwt = np.ascontiguousarray(docs_with_term, np.uint64)
    offs = np.ascontiguousarray(offsets, np.uint32)
    n = lib().orc_score_all_phrase(C.byref(seg), metas.ctypes.data, n_terms, offs.ctypes.data,
                                   C.byref(scorer), boost, docs_with_field, dwt.ctypes.data,
                                   total_term_freq, scores.ctypes.data, pf.ctypes.data)
    if n < 0:
        raise ValueError("orc_score_all_phrase failed")
    return scores, pf


# ---------------------------------------------- term dictionary / columnstore --

def encode_term_metas(metas, has_pos: bool = False, has_pay: bool = False) -> np.ndarray:
    """postings_writer::encode of consecutive terms of ONE dictionary block (oracle's writer
    twin: formats_10.cpp:576-604)."""
    m = _metas_array(metas)
    last = np.zeros(1, m.dtype)
    out = np.zeros(64 * len(m) + 64, np.uint8)
    at = 0
    for i in range(len(m)):
        n = lib().orc_encode_term_meta(m[i:i + 1].ctypes.data, last.ctypes.data, int(has_pos),
                                       int(has_pay), out[at:].ctypes.data, out.size - at)
        assert n > 0
        at += n
    return out[:at].copy()


def decode_term_metas(stream, n: int, has_freq: bool = True, has_pos: bool = False,
                      has_pay: bool = False) -> np.ndarray:
    """postings_reader::decode of `n` consecutive records of one block (formats_10.cpp:3421-3456)."""
    buf = np.ascontiguousarray(stream, np.uint8)
    out = np.zeros(n, TERM_META)
    state = np.zeros(1, out.dtype)
    state["pos_end"] = np.uint64(0xFFFFFFFFFFFFFFFF)
    at = 0
    for i in range(n):
        used = lib().orc_decode_term_meta(buf[at:].ctypes.data, buf.size - at, int(has_freq),
                                          int(has_pos), int(has_pay), state.ctypes.data)
        assert used > 0, (i, used)
        at += used
        out[i] = state[0]
    assert at == buf.size, (at, buf.size)
    return out


def read_document_mask(dm) -> np.ndarray:
    """DocumentMaskReader::read (formats_10.cpp:3275-3312): the deleted doc ids of a `.doc_mask`."""
    b = np.ascontiguousarray(dm, np.uint8)
    n = lib().orc_read_document_mask(b.ctypes.data, b.size, None, 0)
    if n < 0:
        raise ValueError("orc_read_document_mask: corrupt")
    out = np.zeros(max(int(n), 1), np.uint32)
    lib().orc_read_document_mask(b.ctypes.data, b.size, out.ctypes.data, out.size)
    return out[:n]


def walk_term_dictionary(tm, root_start: int, has_freq=True, has_pos=False, has_pay=False):
    """The reference term iterator's walk of `.tm` from the root block -> (terms, metas)."""
    buf = np.ascontiguousarray(tm, np.uint8)
    n, nbytes = C.c_uint32(), C.c_uint64()
    rc = lib().orc_walk_term_dictionary(buf.ctypes.data, buf.size, root_start, int(has_freq),
                                        int(has_pos), int(has_pay), C.byref(n), C.byref(nbytes),
                                        None, None, None)
    if rc != 0:
        raise ValueError("orc_walk_term_dictionary failed: %d" % rc)
    lens = np.zeros(max(n.value, 1), np.uint32)
    blob = np.zeros(max(nbytes.value, 1), np.uint8)
    metas = np.zeros(max(n.value, 1), TERM_META)
    rc = lib().orc_walk_term_dictionary(buf.ctypes.data, buf.size, root_start, int(has_freq),
                                        int(has_pos), int(has_pay), C.byref(n), C.byref(nbytes),
                                        lens.ctypes.data, blob.ctypes.data, metas.ctypes.data)
    assert rc == 0
    terms, at = [], 0
    for i in range(n.value):
        terms.append(bytes(blob[at:at + lens[i]]))
        at += int(lens[i])
    return terms, metas[:n.value]


def read_fixed_column(csi, csd, column_id: int):
    """columnstore2 reader: (value_bytes, min_doc, payload bytes, values[docs * value_bytes])."""
    a = np.ascontiguousarray(csi, np.uint8)
    d = np.ascontiguousarray(csd, np.uint8)
    vb, mn, docs, pl = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    payload = np.zeros(64, np.uint8)
    rc = lib().orc_read_fixed_column(a.ctypes.data, a.size, d.ctypes.data, d.size, column_id,
                                     C.byref(vb), C.byref(mn), C.byref(docs), payload.ctypes.data,
                                     payload.size, C.byref(pl), None, 0)
    if rc != 0:
        raise ValueError("orc_read_fixed_column failed: %d" % rc)
    values = np.zeros(docs.value * vb.value, np.uint8)
    rc = lib().orc_read_fixed_column(a.ctypes.data, a.size, d.ctypes.data, d.size, column_id,
                                     C.byref(vb), C.byref(mn), C.byref(docs), payload.ctypes.data,
                                     payload.size, C.byref(pl), values.ctypes.data, values.size)
    assert rc == 0
    return vb.value, mn.value, bytes(payload[:pl.value]), values


def scored_states(visits_docs_counts, limit: int):
    """limited_sample_collector<term_frequency>::collect (core/search/limited_sample_collector.hpp
    :67-120) driven by multiterm_visitor (:262-300) over the segments' term visits, restated with
    its own explicit heap: scored_states_ + scored_states_heap_ (indices, std::push_heap /
    pop_heap with `comparer_(rhs.key, lhs.key)` — a min-heap on the key), key =
    term_frequency{offset, frequency}: `frequency <, then offset <` (:243-258).
    visits_docs_counts[s] = term_meta::docs_count of the terms segment s's visit yields, in order.
    Returns the (segment, offset) pairs left in scored_states_, sorted."""
    states = []     # scored_states_: [key(frequency, offset), segment]
    heap = []       # scored_states_heap_: indices into states

    def comp(lhs, rhs):     # push() / pop(): comparer_(scored_states_[rhs].key, scored_states_[lhs].key)
        a, b = states[rhs][0], states[lhs][0]
        return a[0] < b[0] or (a[0] == b[0] and a[1] < b[1])      # term_frequency::operator<

    def push_heap_(hole, top, value):     # libstdc++ std::__push_heap
        parent = (hole - 1) // 2
        while hole > top and comp(heap[parent], value):
            heap[hole] = heap[parent]
            hole = parent
            parent = (hole - 1) // 2
        heap[hole] = value

    def adjust_heap(hole, length, value):     # libstdc++ std::__adjust_heap
        top, second = hole, hole
        while second < (length - 1) // 2:
            second = 2 * (second + 1)
            if comp(heap[second], heap[second - 1]):
                second -= 1
            heap[hole] = heap[second]
            hole = second
        if (length & 1) == 0 and second == (length - 2) // 2:
            second = 2 * (second + 1)
            heap[hole] = heap[second - 1]
            hole = second - 1
        push_heap_(hole, top, value)

    for s, counts in enumerate(visits_docs_counts):
        for offset, freq in enumerate(counts):      # multiterm_visitor::visit: ++key_.offset
            key = (int(freq), offset)
            if not limit:
                continue                             # "state will not be scored"
            if len(states) < limit:
                heap.append(len(states))
                states.append([key, s])
                push_heap_(len(heap) - 1, 0, heap[-1])                  # std::push_heap
                continue
            m = heap[0]
            if states[m][0] < key:                   # scored_states_[min_state_idx].key < key
                value = heap[-1]                      # std::pop_heap: the min goes to the back
                heap[-1] = heap[0]
                adjust_heap(0, len(heap) - 1, value)
                states[m] = [key, s]                  # "update min state"
                push_heap_(len(heap) - 1, 0, heap[-1])                  # std::push_heap
    return sorted((s, key[1]) for key, s in states)
