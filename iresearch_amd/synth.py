"""ctypes view of libirs_synth.so — the synthetic IResearch segment builder
(iresearch_amd/index/synth_index.h).  Host only; used by tests and bench.py to
produce `.doc` bytes, term metas, the Norm2 column and field statistics."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _build

LAYOUT_SCALAR = 0  # "1_5"      formats_10.cpp:86-131
LAYOUT_SIMD4 = 1   # "1_5simd"  formats_10.cpp:4122-4157
WAND_MAX_FREQ, WAND_MIN_NORM, WAND_DIV_NORM = 0, 1, 2   # wand_writer.hpp:130-148
SEED = 20260926

# version10::term_meta (formats_10_attributes.hpp:30-50)
TERM_META = np.dtype(
    [("docs_count", "<u4"), ("freq", "<u4"), ("doc_start", "<u8"), ("pos_start", "<u8"),
     ("pos_end", "<u8"), ("pay_start", "<u8"), ("e_skip_start", "<u8")],
    align=True,
)
assert TERM_META.itemsize == 48


class _Params(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("first_doc", C.c_uint64), ("num_docs", C.c_uint32),
        ("vocab_log2", C.c_uint32), ("max_rank", C.c_uint32), ("layout", C.c_uint32),
        ("mean_len", C.c_uint32), ("stddev_len", C.c_uint32), ("threads", C.c_uint32),
        ("keep_postings", C.c_uint32), ("wand_count", C.c_uint32), ("wand_kind", C.c_uint32),
        ("with_positions", C.c_uint32), ("one_based_positions", C.c_uint32),
        ("topic_docs", C.c_uint32), ("topic_percent", C.c_uint32), ("topic_terms", C.c_uint32),
        ("reserved1", C.c_uint32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(str(_build.build_synth()))
        L.irs_synth_build.argtypes = [C.POINTER(_Params), C.POINTER(C.c_void_p)]
        L.irs_synth_build.restype = C.c_int
        L.irs_synth_free.argtypes = [C.c_void_p]
        L.irs_synth_free.restype = None
        for name in ("irs_synth_doc_bytes", "irs_synth_norms", "irs_synth_pos_bytes"):
            f = getattr(L, name)
            f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
            f.restype = C.c_void_p
        L.irs_synth_term_metas.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.irs_synth_term_metas.restype = C.c_void_p
        for name in ("irs_synth_docs_with_field", "irs_synth_total_term_freq"):
            f = getattr(L, name)
            f.argtypes = [C.c_void_p]
            f.restype = C.c_uint64
        L.irs_synth_postings.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
        L.irs_synth_postings.restype = C.c_int
        L.irs_synth_positions.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_uint64)]
        L.irs_synth_positions.restype = C.c_int
        L.irs_synth_encode_term_pos.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                                C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                                C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p,
                                                C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]
        L.irs_synth_encode_term_pos.restype = C.c_int64
        L.irs_synth_encode_term_pos_v.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                                  C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                                  C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64,
                                                  C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                                  C.c_void_p]
        L.irs_synth_encode_term_pos_v.restype = C.c_int64
        L.irs_synth_wrap_file.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.irs_synth_wrap_file.restype = C.c_int64
        L.irs_synth_wrap_pos_file.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                              C.c_uint64, C.POINTER(C.c_uint64)]
        L.irs_synth_wrap_pos_file.restype = C.c_int64
        L.irs_synth_encode_term.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
        L.irs_synth_encode_term.restype = C.c_int64
        L.irs_synth_encode_term_wand.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                                 C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                                 C.c_void_p, C.c_uint64, C.c_void_p]
        L.irs_synth_encode_term_wand.restype = C.c_int64
        L.irs_synth_wrap_doc_file.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                              C.c_uint64, C.POINTER(C.c_uint64)]
        L.irs_synth_wrap_doc_file.restype = C.c_int64
        L.irs_synth_doc_length.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
        L.irs_synth_doc_length.restype = C.c_uint32
        L.irs_synth_queries.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_void_p]
        L.irs_synth_queries.restype = C.c_int
        L.irs_synth_term_meta_stream.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.c_uint32, C.c_void_p, C.c_uint64]
        L.irs_synth_term_meta_stream.restype = C.c_int64
        L.irs_synth_document_mask.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.irs_synth_document_mask.restype = C.c_int64
        L.irs_synth_term_dictionary.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                                C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                C.c_uint32, C.c_void_p, C.c_uint64,
                                                C.POINTER(C.c_uint64)]
        L.irs_synth_term_dictionary.restype = C.c_int64
        L.irs_synth_columnstore.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                            C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint32)]
        L.irs_synth_columnstore.restype = C.c_int64
        _lib = L
    return _lib


def _view(ptr, nbytes, dtype):
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype)


@dataclass
class SynthSegment:
    """One synthetic segment: everything the hot path reads in the reference."""
    doc_file: np.ndarray        # uint8, the whole `.doc` image (header+terms+footer)
    norms: np.ndarray           # uint8, Norm2 of local doc ids 1..N
    metas: np.ndarray           # TERM_META[max_rank]; metas[r-1] = rank r
    docs_with_field: int        # term_reader::docs_count()   (bm25.cpp:54)
    total_term_freq: int        # field frequency attribute    (bm25.cpp:55-57)
    layout: int
    num_docs: int
    postings: dict | None = None  # rank -> (docs u32[], freqs u32[]) when kept
    wand_count: int = 0           # scorers the field was indexed with (wand data in `.doc`)
    pos_file: np.ndarray | None = None   # uint8, the whole `.pos` image (field with POS)
    positions: dict | None = None        # rank -> u32[Σ freqs] positions, doc after doc, when kept
    pos_one_based: bool = False          # formats 1_0..1_2: one-based position storage
    wand_type: int = 0                   # Scorer::WandType of the scorer that wrote the wand data
    doc_mask: np.ndarray | None = None   # uint32 ids of deleted docs (the segment's DocumentMask)

    def meta(self, rank: int) -> np.void:
        return self.metas[rank - 1]


def build_segment(num_docs: int, max_rank: int = 4096, *, layout: int = LAYOUT_SIMD4,
                  seed: int = SEED, first_doc: int = 0, vocab_log2: int = 20,
                  mean_len: int = 100, stddev_len: int = 30, threads: int = 0,
                  keep_postings: bool = False, wand_count: int = 0,
                  wand_kind: int = WAND_MIN_NORM, with_positions: bool = False,
                  one_based_positions: bool = False, topic_docs: int = 0,
                  topic_percent: int = 0, topic_terms: int = 16) -> SynthSegment:
    L = lib()
    p = _Params(seed, first_doc, num_docs, vocab_log2, max_rank, layout, mean_len,
                stddev_len, threads, int(keep_postings), wand_count, wand_kind,
                int(with_positions), int(one_based_positions), int(topic_docs),
                int(topic_percent), int(topic_terms), 0)
    h = C.c_void_p()
    rc = L.irs_synth_build(C.byref(p), C.byref(h))
    if rc != 0:
        raise ValueError("irs_synth_build failed: %d" % rc)
    try:
        n = C.c_uint64()
        ptr = L.irs_synth_doc_bytes(h, C.byref(n))
        doc_file = _view(ptr, n.value, np.uint8).copy()
        ptr = L.irs_synth_norms(h, C.byref(n))
        norms = _view(ptr, n.value, np.uint8).copy()
        m = C.c_uint32()
        ptr = L.irs_synth_term_metas(h, C.byref(m))
        metas = _view(ptr, m.value * TERM_META.itemsize, TERM_META).copy()
        pos_file = positions = None
        if with_positions:
            ptr = L.irs_synth_pos_bytes(h, C.byref(n))
            pos_file = _view(ptr, n.value, np.uint8).copy()
            if keep_postings:
                positions = {}
                for r in range(1, max_rank + 1):
                    pp, c = C.c_void_p(), C.c_uint64()
                    L.irs_synth_positions(h, r, C.byref(pp), C.byref(c))
                    positions[r] = (_view(pp.value, 4 * c.value, np.uint32).copy() if c.value
                                    else np.zeros(0, np.uint32))
        postings = None
        if keep_postings:
            postings = {}
            for r in range(1, max_rank + 1):
                d, f, c = C.c_void_p(), C.c_void_p(), C.c_uint32()
                L.irs_synth_postings(h, r, C.byref(d), C.byref(f), C.byref(c))
                if c.value:
                    postings[r] = (_view(d.value, 4 * c.value, np.uint32).copy(),
                                   _view(f.value, 4 * c.value, np.uint32).copy())
                else:
                    postings[r] = (np.zeros(0, np.uint32), np.zeros(0, np.uint32))
        return SynthSegment(doc_file, norms, metas, L.irs_synth_docs_with_field(h),
                            L.irs_synth_total_term_freq(h), layout, num_docs, postings,
                            wand_count, pos_file, positions, bool(one_based_positions),
                            # WAND_* (the writer's tags) -> Scorer::WandType (scorer.hpp:196-201)
                            {WAND_MAX_FREQ: 2, WAND_MIN_NORM: 3, WAND_DIV_NORM: 1}[wand_kind]
                            if wand_count else 0)
    finally:
        L.irs_synth_free(h)


def encode_term_pos(docs, freqs, positions, segment_docs: int, layout: int = LAYOUT_SIMD4,
                    norms=None, wand_kinds=(), one_based: bool = False):
    """postings_writer::write for one list of a field with POS -> (doc bytes, pos bytes, meta);
    positions = Σ freqs values, doc after doc (ascending and >= 1 within a doc)."""
    docs = np.ascontiguousarray(docs, dtype=np.uint32)
    freqs = np.ascontiguousarray(freqs, dtype=np.uint32)
    positions = np.ascontiguousarray(positions, dtype=np.uint32)
    assert docs.shape == freqs.shape and positions.size == int(freqs.sum())
    kinds = np.ascontiguousarray(wand_kinds, dtype=np.uint32)
    cap = 64 + 12 * len(docs) + 1024 + (16 * len(kinds) + 16) * (len(docs) // 128 + 4)
    out = np.zeros(cap, np.uint8)
    pcap = 64 + 5 * positions.size
    pout = np.zeros(pcap, np.uint8)
    plen = C.c_uint64()
    meta = np.zeros(1, TERM_META)
    nrm = None if norms is None else np.ascontiguousarray(norms, np.uint8)
    n = lib().irs_synth_encode_term_pos_v(docs.ctypes.data, freqs.ctypes.data,
                                          positions.ctypes.data, len(docs), segment_docs, layout,
                                          None if nrm is None else nrm.ctypes.data,
                                          kinds.ctypes.data if kinds.size else None, kinds.size,
                                          int(one_based), out.ctypes.data, cap, pout.ctypes.data,
                                          pcap, C.byref(plen), meta.ctypes.data)
    if n < 0:
        raise ValueError("irs_synth_encode_term_pos failed: %d" % n)
    return out[:n].copy(), pout[:plen.value].copy(), meta[0]


def encode_term(docs, freqs, segment_docs: int, layout: int = LAYOUT_SIMD4, norms=None,
                wand_kinds=()):
    """postings_writer::write for one explicit list -> (bytes, meta).  wand_kinds: one WAND_*
    per scorer the field is indexed with (norms = 1-byte Norm2 column, norms[doc - 1])."""
    docs = np.ascontiguousarray(docs, dtype=np.uint32)
    if freqs is not None:  # None: a field without IndexFeatures::FREQ
        freqs = np.ascontiguousarray(freqs, dtype=np.uint32)
        assert docs.shape == freqs.shape
    cap = 64 + 12 * len(docs) + 1024
    out = np.zeros(cap, np.uint8)
    meta = np.zeros(1, TERM_META)
    kinds = np.ascontiguousarray(wand_kinds, dtype=np.uint32)
    cap += 16 * len(kinds) * (len(docs) // 128 + 4)
    out = np.zeros(cap, np.uint8)
    nrm = None if norms is None else np.ascontiguousarray(norms, np.uint8)
    n = lib().irs_synth_encode_term_wand(docs.ctypes.data,
                                         None if freqs is None else freqs.ctypes.data, len(docs),
                                         segment_docs, layout,
                                         None if nrm is None else nrm.ctypes.data,
                                         kinds.ctypes.data if kinds.size else None, kinds.size,
                                         out.ctypes.data, cap, meta.ctypes.data)
    if n < 0:
        raise ValueError("irs_synth_encode_term failed: %d" % n)
    return out[:n].copy(), meta[0]


def segment_from_lists(lists, num_docs: int, layout: int = LAYOUT_SIMD4, norms=None,
                       wand_kinds=(), one_based: bool = False):
    """Build a `.doc` image from explicit [(docs, freqs), ...] posting lists."""
    if len(wand_kinds) and (norms is None or norms is False) and \
            any(k != WAND_MAX_FREQ for k in wand_kinds):
        raise ValueError("MIN_NORM / DIV_NORM wand data needs the norm column")
    body = []
    pbody = []
    metas = np.zeros(len(lists), TERM_META)
    off = poff = 0
    with_pos = bool(lists) and len(lists[0]) == 3   # [(docs, freqs, positions), ...]: field with POS
    for i, entry in enumerate(lists):
        wn = norms if len(wand_kinds) and norms is not None and norms is not False else None
        if with_pos:
            b, pb, m = encode_term_pos(entry[0], entry[1], entry[2], num_docs, layout, wn,
                                       wand_kinds, one_based)
            metas[i] = m
            metas[i]["pos_start"] = poff
            poff += len(pb)
            pbody.append(pb)
        else:
            b, m = encode_term(entry[0], entry[1], num_docs, layout, wn, wand_kinds)
            metas[i] = m
        metas[i]["doc_start"] = off
        off += len(b)
        body.append(b)
    body = np.concatenate(body) if body else np.zeros(0, np.uint8)
    pos_file = None
    if with_pos:
        pbody = np.concatenate(pbody) if pbody else np.zeros(0, np.uint8)
        pout = np.zeros(len(pbody) + 128, np.uint8)
        phdr = C.c_uint64()
        pn = lib().irs_synth_wrap_file(pbody.ctypes.data, len(pbody), layout, 1, int(one_based),
                                       pout.ctypes.data, len(pout), C.byref(phdr))
        if pn < 0:
            raise ValueError("irs_synth_wrap_pos_file failed")
        metas["pos_start"] += phdr.value
        pos_file = pout[:pn].copy()
    out = np.zeros(len(body) + 128, np.uint8)
    hdr = C.c_uint64()
    n = lib().irs_synth_wrap_file(body.ctypes.data, len(body), layout, 0, int(one_based),
                                  out.ctypes.data, len(out), C.byref(hdr))
    if n < 0:
        raise ValueError("irs_synth_wrap_doc_file failed")
    metas["doc_start"] += hdr.value
    # Scorer::WandType (scorer.hpp:196-201) of the scorer that wrote slot 0 of the wand data
    wand_type = {WAND_MAX_FREQ: 2, WAND_MIN_NORM: 3, WAND_DIV_NORM: 1}[wand_kinds[0]] \
        if len(wand_kinds) else 0
    if norms is False:  # no Norm2 column at all
        return SynthSegment(out[:n].copy(), None, metas, num_docs, num_docs, layout, num_docs,
                            None, len(wand_kinds), pos_file, None, bool(one_based), wand_type)
    if norms is None:
        norms = np.ones(num_docs, np.uint8)
    ttf = int(np.asarray(norms, dtype=np.uint64).sum())
    return SynthSegment(out[:n].copy(), np.ascontiguousarray(norms, np.uint8), metas,
                        num_docs, ttf, layout, num_docs, None, len(wand_kinds), pos_file, None,
                        bool(one_based), wand_type)


def make_queries(n_queries: int, n_terms: int, lo_rank: int = 16, hi_rank: int = 4096,
                 seed: int = SEED + 2) -> np.ndarray:
    out = np.zeros((n_queries, n_terms), np.uint32)
    rc = lib().irs_synth_queries(seed, n_queries, n_terms, lo_rank, hi_rank, out.ctypes.data)
    if rc != 0:
        raise ValueError("irs_synth_queries failed")
    return out


def doc_length(global_doc: int, seed: int = SEED, mean_len: int = 100, stddev_len: int = 30):
    return lib().irs_synth_doc_length(seed, global_doc, mean_len, stddev_len)


# -------------------------------------------- term dictionary / columnstore files --

def term_meta_stream(metas, has_freq=True, has_pos=False, has_pay=False) -> np.ndarray:
    """The stats records of ONE dictionary block (postings_writer::encode, formats_10.cpp:576-604)."""
    m = np.ascontiguousarray(metas, TERM_META)
    out = np.zeros(64 * len(m) + 64, np.uint8)
    n = lib().irs_synth_term_meta_stream(m.ctypes.data, len(m), int(has_freq), int(has_pos),
                                         int(has_pay), out.ctypes.data, out.size)
    if n < 0:
        raise ValueError("irs_synth_term_meta_stream failed: %d" % n)
    return out[:n].copy()


def term_bytes_of(ordinal: int) -> bytes:
    """The synthetic term of ordinal i (rank i + 1): its 4-byte big-endian number — ascending
    bytewise like the ordinals, with long shared prefixes (a deep block tree)."""
    return int(ordinal).to_bytes(4, "big")


def term_dictionary(terms, metas, has_freq=True, has_pos=False, has_pay=False, min_block=25,
                    max_block=48):
    """`.tm` of one field (field_writer, formats_burst_trie.cpp:1023-1196) -> (bytes, root_start)."""
    m = np.ascontiguousarray(metas, TERM_META)
    assert len(terms) == len(m)
    blob = np.frombuffer(b"".join(terms), np.uint8) if terms else np.zeros(0, np.uint8)
    blob = np.ascontiguousarray(blob)
    lens = np.array([len(t) for t in terms], np.uint32)
    out = np.zeros(blob.size + 64 * len(m) + 4096, np.uint8)
    root = C.c_uint64()
    n = lib().irs_synth_term_dictionary(blob.ctypes.data if blob.size else None,
                                        lens.ctypes.data if lens.size else None,
                                        m.ctypes.data if len(m) else None, len(m), int(has_freq),
                                        int(has_pos), int(has_pay), min_block, max_block,
                                        out.ctypes.data, out.size, C.byref(root))
    if n < 0:
        raise ValueError("irs_synth_term_dictionary failed: %d" % n)
    return out[:n].copy(), int(root.value)


def document_mask(docs) -> np.ndarray:
    """`.doc_mask` (DocumentMaskWriter::write, formats_10.cpp:3245-3268) of the deleted doc ids."""
    d = np.ascontiguousarray(docs, np.uint32)
    out = np.zeros(5 * d.size + 128, np.uint8)
    n = lib().irs_synth_document_mask(d.ctypes.data if d.size else None, d.size, out.ctypes.data, out.size)
    if n < 0:
        raise ValueError("irs_synth_document_mask failed: %d" % n)
    return out[:n].copy()


def norm2_header(width: int, lo: int, hi: int) -> bytes:
    """Norm2Header::Write (norm.cpp:107-115): version 0, bytes per value, min and max length."""
    return bytes([0, width]) + int(lo).to_bytes(4, "big") + int(hi).to_bytes(4, "big")


def columnstore(values, width: int, min_doc: int = 1, payload: bytes = b"", dense_fixed=False,
                lead_columns: int = 2):
    """columnstore2 `.csd` / `.csi` with one anonymous fixed-length column -> (csd, csi, id)."""
    v = np.ascontiguousarray(values, np.uint8)
    n_docs = v.size // width
    csd = np.zeros(v.size + 4096 + 8 * (n_docs // 65536 + 2), np.uint8)
    csi = np.zeros(4096 + 8 * (n_docs // 65536 + 2) + 64 * lead_columns, np.uint8)
    pl = np.frombuffer(payload, np.uint8).copy() if payload else np.zeros(1, np.uint8)
    dl, il, cid = C.c_uint64(), C.c_uint64(), C.c_uint32()
    rc = lib().irs_synth_columnstore(v.ctypes.data, width, n_docs, min_doc, pl.ctypes.data,
                                     len(payload), int(dense_fixed), lead_columns,
                                     csd.ctypes.data, csd.size, C.byref(dl), csi.ctypes.data,
                                     csi.size, C.byref(il), C.byref(cid))
    if rc != 0:
        raise ValueError("irs_synth_columnstore failed: %d" % rc)
    return csd[:dl.value].copy(), csi[:il.value].copy(), int(cid.value)
