"""Host-side mirror of the reference interfaces on the hot path, above the C ABI.

Names follow IResearch: a *segment* holds posting lists of *terms*; a *filter*
(`by_term`, `Or`, `And`) is `prepare`d against an *index* (all segments) to
collect statistics, then executed per segment; a *scorer* (`BM25`, `TFIDF`)
turns statistics into per-term score functions.

  irs::BM25::collect          core/search/bm25.cpp:366-410     -> BM25.collect
  irs::TFIDF::collect         core/search/tfidf.cpp:263-278    -> TFIDF.collect
  by_term::prepare            core/search/term_filter.cpp:92-129 -> prepare()
  by_phrase (FixedPrepareCollect) core/search/phrase_filter.cpp:212-293 -> prepare()
  filter::prepared::execute   core/search/filter.hpp:52-78     -> SegmentReader.execute()
  utils/index-search.cpp:719-787 (heap over all segments)     -> Index.search()

This module holds no posting decode or scoring arithmetic of its own beyond the
per-term statistics (which the reference also computes on the CPU once per
query): everything per posting happens in libirs_hip.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import (HIT, NO_TERM, OP_AND, OP_MINMATCH, OP_OR, OP_PHRASE, QUERY, SCORE_BM1, SCORE_BM15, SCORE_BM25,
                   SCORE_TFIDF, SCORE_TFIDF_NORM, TERM_META, TERM_SCORER, SegmentDesc)

f32 = np.float32


# ------------------------------------------------------------------ scorers --

@dataclass(frozen=True)
class TermStats:
    """BM25Stats minus the cache (bm25.hpp:48-57) / TFIDFStats."""
    idf: np.float32
    norm_const: np.float32 = f32(0)
    norm_length: np.float32 = f32(0)


class BM25:
    """irs::BM25 (bm25.hpp:59-117): k = 1.2, b = 0.75 by default."""

    def __init__(self, k: float = 1.2, b: float = 0.75):
        self.k, self.b = f32(k), f32(b)

    def collect(self, docs_with_field: int, docs_with_term: int, total_term_freq: int):
        # bm25.cpp:381-383 — double log1p, then cast
        idf = f32(math.log1p((float(docs_with_field - docs_with_term) + 0.5) /
                             (float(docs_with_term) + 0.5)))
        if self.k == 0 or self.b == 0:  # !NeedsNorm() bm25.cpp:387-390
            return TermStats(idf, self.k, f32(0))
        kb = f32(self.k * self.b)
        norm_const = f32(self.k - kb)
        if total_term_freq and docs_with_field:
            avg_dl = f32(f32(total_term_freq) / f32(docs_with_field))
            norm_length = f32(kb / avg_dl)
        else:
            norm_length = kb
        return TermStats(idf, norm_const, norm_length)

    def term_scorer(self, st: TermStats, boost: float = 1.0):
        # BM1Context: num = boost * (k + 1) * idf (bm25.cpp:201)
        c0 = f32(f32(f32(boost) * f32(self.k + f32(1))) * st.idf)
        if self.k == 0:
            kind = SCORE_BM1
        elif self.b == 0:
            kind = SCORE_BM15
        else:
            kind = SCORE_BM25
        return kind, c0, st.norm_const, st.norm_length


class TFIDF:
    """irs::TFIDF (tfidf.hpp): with_norms == normalize()."""

    def __init__(self, with_norms: bool = False):
        self.with_norms = bool(with_norms)

    def collect(self, docs_with_field: int, docs_with_term: int, total_term_freq: int):
        return TermStats(f32(math.log1p((docs_with_field + 1.0) / (docs_with_term + 1.0))))

    def term_scorer(self, st: TermStats, boost: float = 1.0):
        c0 = f32(f32(boost) * st.idf)  # TFIDFContext: idf = boost * idf.value
        return (SCORE_TFIDF_NORM if self.with_norms else SCORE_TFIDF), c0, f32(0), f32(0)


# ------------------------------------------------------------------ filters --

@dataclass
class by_term:
    """irs::by_term on the benchmark field; `term` is the ordinal in the term table."""
    term: int
    boost: float = 1.0


# irs::ScoreMergeType (scorer.hpp:224-236) of a boolean filter: boolean_filter::merge_type()
MERGE_SUM, MERGE_MAX, MERGE_MIN = 0, 1, 2


@dataclass
class Or:
    """irs::Or; min_match > 1 is Or::min_match_count() (boolean_filter.hpp); `merge` its
    merge_type()."""
    subs: list
    min_match: int = 1
    merge: int = MERGE_SUM

    @property
    def op(self):
        return OP_MINMATCH if self.min_match > 1 else OP_OR


@dataclass
class And:
    subs: list
    op: int = OP_AND
    min_match: int = 0
    merge: int = MERGE_SUM


@dataclass
class by_phrase:
    """irs::by_phrase of plain terms (by_phrase_options::push_back<by_term_options>):
    `terms` are term ordinals; `offsets[i]` = position of term i relative to the first
    (default: consecutive words)."""
    terms: list
    offsets: list | None = None
    boost: float = 1.0

    def __post_init__(self):
        if self.offsets is None:
            self.offsets = list(range(len(self.terms)))
        if len(self.offsets) != len(self.terms) or (self.offsets and self.offsets[0] != 0):
            raise ValueError("offsets are relative to the first term")


def _terms_of(flt):
    if isinstance(flt, by_term):
        return OP_OR, [flt]
    if not flt.subs or any(not isinstance(s, by_term) for s in flt.subs):
        raise ValueError("only flat Or/And of by_term are on the GPU path")
    return flt.op, list(flt.subs)


@dataclass
class PreparedQuery:
    """filter::prepared: the op plus (term, global stats, boost) per sub-filter."""
    op: int
    terms: list            # term ordinals
    scorers: list          # (kind, c0, norm_const, norm_length) per term
    min_match: int = 0
    offsets: list | None = None   # OP_PHRASE: position of every term in the phrase
    merge: int = MERGE_SUM


# ------------------------------------------------------------------ segment --

class SegmentReader:
    """irs::SubReader + postings_reader of one segment, resident on one GPU."""

    def __init__(self, doc_file, metas, num_docs, layout, norms=None, norm_width=1,
                 docs_with_field=None, total_term_freq=0, device=0, has_freq=True, L=None,
                 wand_count=0, pos_file=None, pos_features=0, norm_kind=0, wand_type=0,
                 doc_mask=None):
        self.L = L or _lib.lib()
        self.doc_file = np.ascontiguousarray(doc_file, np.uint8)
        self.metas = np.zeros(len(metas), TERM_META)
        for name in TERM_META.names:
            self.metas[name] = np.asarray(metas)[name]
        self.num_docs, self.layout, self.device = int(num_docs), int(layout), int(device)
        self.norms = None if norms is None else np.ascontiguousarray(norms, np.uint8)
        self.norm_width = norm_width
        self.docs_with_field = int(num_docs if docs_with_field is None else docs_with_field)
        self.total_term_freq = int(total_term_freq)
        self.pos_file = None if pos_file is None else np.ascontiguousarray(pos_file, np.uint8)
        # the segment's DocumentMask (deleted doc ids): SegmentReaderImpl::mask, applied by every batch
        self.doc_mask = None if doc_mask is None else np.ascontiguousarray(doc_mask, np.uint32)
        desc = SegmentDesc(
            device, layout, self.doc_file.ctypes.data, self.doc_file.size, num_docs,
            int(has_freq), None if self.norms is None else self.norms.ctypes.data, norm_width, 1,
            0 if self.norms is None else self.norms.size // norm_width,
            self.metas.ctypes.data, len(self.metas), int(wand_count),
            None if self.pos_file is None else self.pos_file.ctypes.data,
            0 if self.pos_file is None else self.pos_file.size, int(pos_features), int(norm_kind),
            int(wand_type),
            None if self.doc_mask is None or not self.doc_mask.size else self.doc_mask.ctypes.data,
            0 if self.doc_mask is None else self.doc_mask.size)
        h = C.c_void_p()
        _lib.check(self.L, self.L.irs_hip_segment_open(C.byref(desc), C.byref(h)),
                   "irs_hip_segment_open")
        self.handle = h

    @classmethod
    def from_synth(cls, seg, device=0, L=None, has_freq=True, doc_mask=None):
        return cls(seg.doc_file, seg.metas, seg.num_docs, seg.layout, seg.norms, 1,
                   seg.docs_with_field, seg.total_term_freq, device, has_freq, L,
                   getattr(seg, "wand_count", 0), getattr(seg, "pos_file", None),
                   wand_type=getattr(seg, "wand_type", 0),
                   doc_mask=getattr(seg, "doc_mask", None) if doc_mask is None else doc_mask)

    def close(self):
        if self.handle:
            self.L.irs_hip_segment_close(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_bytes(self) -> int:
        return self.L.irs_hip_segment_device_bytes(self.handle)

    def live_docs_count(self) -> int:
        """SubReader::live_docs_count: docs that are not in the segment's doc_mask."""
        return self.L.irs_hip_segment_live_docs(self.handle)

    def decode_term(self, term: int, want_freq: bool = True):
        n = int(self.metas[term]["docs_count"])
        docs = np.zeros(max(n, 1), np.uint32)
        freqs = np.zeros(max(n, 1), np.uint32) if want_freq else None
        cnt = C.c_uint32()
        _lib.check(self.L, self.L.irs_hip_decode_term(
            self.handle, term, docs.ctypes.data, None if freqs is None else freqs.ctypes.data,
            docs.size, C.byref(cnt)), "irs_hip_decode_term")
        return docs[:cnt.value], (None if freqs is None else freqs[:cnt.value])

    def decode_positions(self, term: int):
        """Every position of every doc of the term, doc after doc (term_meta::freq values)."""
        n = int(self.metas[term]["freq"])
        out = np.zeros(max(n, 1), np.uint32)
        cnt = C.c_uint64()
        _lib.check(self.L, self.L.irs_hip_decode_positions(
            self.handle, term, out.ctypes.data, out.size, C.byref(cnt)),
            "irs_hip_decode_positions")
        return out[:cnt.value]

    def bit_union(self, terms, n_words: int, initial=None):
        """postings_reader::bit_union: (bitset as uint64 words, sum of docs_count)."""
        t = np.ascontiguousarray(terms, np.uint32)
        bits = np.zeros(n_words, np.uint64) if initial is None else \
            np.ascontiguousarray(initial, np.uint64).copy()
        cnt = C.c_uint64()
        _lib.check(self.L, self.L.irs_hip_bit_union(
            self.handle, t.ctypes.data if t.size else None, t.size, bits.ctypes.data, n_words,
            C.byref(cnt)), "irs_hip_bit_union")
        return bits, cnt.value

    def bit_union_counts(self, term_sets):
        """The populations of several term sets' unions (irs_hip_bit_union_counts): one number per
        set comes back, the bitsets stay on the device."""
        sets = [np.ascontiguousarray(t, np.uint32) for t in term_sets]
        offsets = np.zeros(len(sets) + 1, np.uint32)
        np.cumsum([len(t) for t in sets], out=offsets[1:])
        flat = np.concatenate(sets) if sets and offsets[-1] else np.zeros(1, np.uint32)
        counts = np.zeros(max(len(sets), 1), np.uint64)
        _lib.check(self.L, self.L.irs_hip_bit_union_counts(
            self.handle, flat.ctypes.data, offsets.ctypes.data, len(sets), counts.ctypes.data),
            "irs_hip_bit_union_counts")
        return counts[:len(sets)]

    def term_directory(self, term: int):
        nb = int(self.metas[term]["docs_count"]) // 128
        last = np.zeros(max(nb, 1), np.uint32)
        offs = np.zeros(max(nb, 1), np.uint64)
        cnt = C.c_uint32()
        _lib.check(self.L, self.L.irs_hip_term_directory(
            self.handle, term, last.ctypes.data, offs.ctypes.data, last.size, C.byref(cnt)),
            "irs_hip_term_directory")
        return last[:cnt.value], offs[:cnt.value]

    def wand_source(self):
        """(blocks whose (max freq, min norm) was read from the index's own wand data, all blocks)."""
        a, b = C.c_uint64(), C.c_uint64()
        _lib.check(self.L, self.L.irs_hip_segment_wand_source(self.handle, C.byref(a), C.byref(b)),
                   "irs_hip_segment_wand_source")
        return a.value, b.value

    def term_blockmax(self, term: int):
        """Per full block of the term: (largest frequency, smallest non-zero norm) — the
        block-max data WAND batches prune with."""
        nb = int(self.metas[term]["docs_count"]) // 128
        mf = np.zeros(max(nb, 1), np.uint32)
        mn = np.zeros(max(nb, 1), np.uint32)
        cnt = C.c_uint32()
        _lib.check(self.L, self.L.irs_hip_term_blockmax(
            self.handle, term, mf.ctypes.data, mn.ctypes.data, mf.size, C.byref(cnt)),
            "irs_hip_term_blockmax")
        return mf[:cnt.value], mn[:cnt.value]

    def batch(self, prepared, k: int):
        return QueryBatch(self, prepared, k)


class QueryArrays:
    """The C arrays irs_hip_batch_create_multi takes for a list of prepared queries on a list
    of segments: irs_hip_query[nq] and irs_hip_term_scorer[n_segs][n_entries] (every segment
    gets the same scorer constants and its own term ordinals).  Building them is the host half
    of filter::prepare; creating a batch from them is one C call."""

    def __init__(self, n_segs, queries, terms, k):
        self.n_segs, self.queries, self.terms, self.k = int(n_segs), queries, terms, int(k)

    @classmethod
    def from_prepared(cls, segs, prepared, k):
        n_entries = sum(len(p.terms) for p in prepared)
        queries = np.zeros(len(prepared), QUERY)
        terms = np.zeros((len(segs), max(n_entries, 1)), TERM_SCORER)
        at = 0
        for q, p in enumerate(prepared):
            queries[q] = (p.op, len(p.terms), at, int(k), p.min_match, p.merge)
            offs = p.offsets if p.offsets is not None else [0] * len(p.terms)
            for t, (kind, c0, nc, nl), off in zip(p.terms, p.scorers, offs):
                for s, sr in enumerate(segs):       # same scorer, the segment's own ordinal
                    present = t is not None and 0 <= t < len(sr.metas)
                    terms[s, at] = (t if present else NO_TERM, kind, c0, nc, nl, off)
                at += 1
        return cls(len(segs), queries, terms, k)


def prepare_disjunctions(term_rows, scorer, segment_stats, segs, k, boost=1.0):
    """prepare() + QueryArrays.from_prepared for the common case — every row of `term_rows`
    (int array [nq][n_terms] of term ordinals) is an Or of by_term filters with one boost —
    with the statistics of ALL terms computed at once (numpy), value for value what
    BM25.collect / TFIDF.collect / term_scorer compute one term at a time (the test suite
    compares the two).  This is the per-batch host work of a serving loop."""
    rows = np.ascontiguousarray(term_rows, np.int64)
    nq, nt = rows.shape
    flat = rows.reshape(-1)
    dwf = sum(s.docs_with_field for s in segment_stats)
    ttf = sum(s.total_term_freq for s in segment_stats)
    dwt = np.zeros(flat.size, np.float64)
    for st in segment_stats:
        dc = np.asarray(st.docs_count)
        ok = (flat >= 0) & (flat < len(dc))
        dwt[ok] += dc[flat[ok]]
    if isinstance(scorer, BM25):
        idf = np.log1p((float(dwf) - dwt + 0.5) / (dwt + 0.5)).astype(np.float32)
        c0 = (f32(f32(boost) * f32(scorer.k + f32(1))) * idf).astype(np.float32)
        probe = scorer.collect(dwf, 1, ttf)
        kind = scorer.term_scorer(probe)[0]
        nc, nl = probe.norm_const, probe.norm_length
    else:
        idf = np.log1p((dwf + 1.0) / (dwt + 1.0)).astype(np.float32)
        c0 = (f32(boost) * idf).astype(np.float32)
        kind, nc, nl = scorer.term_scorer(TermStats(f32(1)))[0], f32(0), f32(0)
    queries = np.zeros(nq, QUERY)
    queries["op"], queries["n_terms"], queries["k"] = OP_OR, nt, int(k)
    queries["first_term"] = np.arange(nq, dtype=np.uint32) * nt
    queries["min_match"], queries["merge"] = 1, MERGE_SUM
    terms = np.zeros((len(segs), max(flat.size, 1)), TERM_SCORER)
    for s, sr in enumerate(segs):
        present = (flat >= 0) & (flat < len(sr.metas))
        terms["term"][s, :flat.size] = np.where(present, flat, NO_TERM).astype(np.uint32)
    terms["kind"][:, :flat.size] = kind
    terms["c0"][:, :flat.size] = c0
    terms["norm_const"][:, :flat.size] = nc
    terms["norm_length"][:, :flat.size] = nl
    return QueryArrays(len(segs), queries, terms, k)


def prepare_filters(filters, scorer, segment_stats, segs, k):
    """prepare() + QueryArrays.from_prepared for ANY list of the flat filters of this path
    (by_term, Or, Or with min_match, And, by_phrase — one list may mix the boolean kinds; phrases
    go in batches of their own) with the statistics of all terms computed at once: value for value
    what the term-by-term prepare() yields (tests/test_host_prepare.py compares the two).  The
    reference harness builds and prepares every task's filter inside its timer
    (index-search.cpp:694-722, in C++); this is that stage for a batch without a Python loop over
    the terms."""
    nq = len(filters)
    ops, nts, mms, mgs = (np.zeros(nq, np.int64) for _ in range(4))
    tl, bl, ol = [], [], []
    ol_any = False
    # (one pass over the filter objects, the inner loops as list comprehensions: for 1000 filters
    # of 8 terms this loop IS the stage's time)
    for q, flt in enumerate(filters):
        cls = type(flt)
        if cls is by_term:
            ops[q], nts[q] = OP_OR, 1
            mgs[q] = MERGE_SUM
            tl.append(flt.term)
            bl.append(flt.boost)
            continue
        if isinstance(flt, by_phrase):
            n = len(flt.terms)
            ops[q], nts[q] = OP_PHRASE, n
            if not ol_any:
                ol = [0] * len(tl)
                ol_any = True
            tl.extend(flt.terms)
            bl.extend([flt.boost] * n)
            ol.extend(flt.offsets)
            continue
        if cls is Or or cls is And:
            subs = flt.subs
            try:
                terms_q = [s.term for s in subs]
                boosts_q = [s.boost for s in subs]
            except AttributeError:
                terms_q = None
            if not subs or terms_q is None or (set(map(type, subs)) != {by_term} and
                                               any(not isinstance(s, by_term) for s in subs)):
                raise ValueError("only flat Or/And of by_term are on the GPU path")
            op = flt.op
        else:
            op, subs = _terms_of(flt)
            terms_q = [s.term for s in subs]
            boosts_q = [s.boost for s in subs]
        ops[q], nts[q] = op, len(terms_q)
        mms[q] = int(getattr(flt, "min_match", 0))
        mgs[q] = int(getattr(flt, "merge", MERGE_SUM))
        tl += terms_q
        bl += boosts_q
        if ol_any:
            ol += [0] * len(terms_q)
    if not ol_any:
        ol = np.zeros(len(tl), np.uint32)
    flat = np.asarray(tl, np.int64)
    boost = np.asarray(bl, np.float32)
    first = np.zeros(nq, np.int64)
    np.cumsum(nts[:-1], out=first[1:])
    dwf = sum(s.docs_with_field for s in segment_stats)
    ttf = sum(s.total_term_freq for s in segment_stats)
    dwt = np.zeros(flat.size, np.float64)
    for st in segment_stats:
        dc = np.asarray(st.docs_count)
        ok = (flat >= 0) & (flat < len(dc))
        dwt[ok] += dc[flat[ok]]
    if isinstance(scorer, BM25):
        idf = np.log1p((float(dwf) - dwt + 0.5) / (dwt + 0.5)).astype(np.float32)
        probe = scorer.collect(dwf, 1, ttf)
        kind = scorer.term_scorer(probe)[0]
        nc, nl = probe.norm_const, probe.norm_length
    else:
        idf = np.log1p((dwf + 1.0) / (dwt + 1.0)).astype(np.float32)
        kind, nc, nl = scorer.term_scorer(TermStats(f32(1)))[0], f32(0), f32(0)
    # by_phrase: ONE stats blob per phrase, into which every term's idf was added in phrase order
    # (float32 `idf +=`: bm25.cpp:381-383, tfidf.cpp:272-275) — every entry carries that sum
    ph = np.nonzero(ops == OP_PHRASE)[0]
    if ph.size:
        acc = np.zeros(ph.size, np.float32)
        for j in range(int(nts[ph].max())):
            on = nts[ph] > j
            acc[on] = (acc[on] + idf[first[ph[on]] + j]).astype(np.float32)
        of_q = np.repeat(np.arange(nq), nts)
        slot = np.full(nq, -1, np.int64)
        slot[ph] = np.arange(ph.size)
        is_ph = slot[of_q] >= 0
        idf = idf.copy()
        idf[is_ph] = acc[slot[of_q[is_ph]]]
    if isinstance(scorer, BM25):
        c0 = ((boost * f32(scorer.k + f32(1))).astype(np.float32) * idf).astype(np.float32)
    else:
        c0 = (boost * idf).astype(np.float32)
    queries = np.zeros(nq, QUERY)
    queries["op"], queries["n_terms"], queries["first_term"] = ops, nts, first
    queries["k"], queries["min_match"], queries["merge"] = int(k), mms, mgs
    terms = np.zeros((len(segs), max(flat.size, 1)), TERM_SCORER)
    for s, sr in enumerate(segs):
        present = (flat >= 0) & (flat < len(sr.metas))
        terms["term"][s, :flat.size] = np.where(present, flat, NO_TERM).astype(np.uint32)
    terms["kind"][:, :flat.size] = kind
    terms["c0"][:, :flat.size] = c0
    terms["norm_const"][:, :flat.size] = nc
    terms["norm_length"][:, :flat.size] = nl
    terms["phrase_offset"][:, :flat.size] = np.asarray(ol, np.uint32)
    return QueryArrays(len(segs), queries, terms, k)


class QueryBatch:
    """A batch of prepared queries on one segment — or on several segments of one device
    at once (irs_hip_batch_create_multi): then every result array gets a leading segment
    axis, [n_segs][nq]..., in the order the readers were given.  `prepared`: a list of
    PreparedQuery, or the QueryArrays made from one."""

    def __init__(self, seg, prepared, k: int | None = None):
        self.segs = list(seg) if isinstance(seg, (list, tuple)) else [seg]
        self.multi = isinstance(seg, (list, tuple))
        arrays = prepared if isinstance(prepared, QueryArrays) else \
            QueryArrays.from_prepared(self.segs, prepared, k)
        assert arrays.n_segs == len(self.segs)
        self.seg, self.L, self.k = self.segs[0], self.segs[0].L, arrays.k
        self.nq_user = len(arrays.queries)
        self.nq = self.nq_user * len(self.segs)          # execution units
        self.queries, self.terms = arrays.queries, arrays.terms
        h = C.c_void_p()
        handles = (C.c_void_p * len(self.segs))(*[sr.handle for sr in self.segs])
        _lib.check(self.L, self.L.irs_hip_batch_create_multi(
            handles, len(self.segs), self.queries.ctypes.data, self.nq_user,
            self.terms.ctypes.data, self.terms.shape[1], C.byref(h)),
            "irs_hip_batch_create_multi")
        self.handle = h

    def configure(self, tile_docs=0, pilot_stride=0, cand_cap=0):
        _lib.check(self.L, self.L.irs_hip_batch_configure(self.handle, tile_docs, pilot_stride,
                                                          cand_cap), "irs_hip_batch_configure")
        return self

    def set_shared_threshold(self, enable=True):
        """One threshold per query for its units on the batch's segments
        (irs_hip_batch_set_shared_threshold): for callers that merge the per-segment lists."""
        _lib.check(self.L, self.L.irs_hip_batch_set_shared_threshold(self.handle, int(bool(enable))),
                   "irs_hip_batch_set_shared_threshold")
        return self

    def set_comm(self, comm):
        """One threshold per query across RANKS (irs_hip_batch_set_comm): `comm` = a
        distributed.Communicator (or None to detach); every rank attaches one to its batch of the
        same queries and runs the batches in the same order."""
        self._comm = comm   # (must outlive the batch)
        _lib.check(self.L, self.L.irs_hip_batch_set_comm(self.handle, comm.handle if comm else None),
                   "irs_hip_batch_set_comm")
        return self

    def set_path(self, path):
        """PATH_AUTO / PATH_ITEMS / PATH_JOINED (irs_hip_batch_set_path)."""
        _lib.check(self.L, self.L.irs_hip_batch_set_path(self.handle, int(path)),
                   "irs_hip_batch_set_path")
        return self

    def set_async(self, enable=True):
        """Hand the host half of run() to the library's worker thread (1), keep it on the caller's
        thread (0), or follow the process default (-1): irs_hip_batch_set_async."""
        _lib.check(self.L, self.L.irs_hip_batch_set_async(self.handle, int(enable)),
                   "irs_hip_batch_set_async")
        return self

    def path(self):
        """Which path the last run took (PATH_ITEMS / PATH_JOINED)."""
        v = C.c_int(0)
        _lib.check(self.L, self.L.irs_hip_batch_path(self.handle, C.byref(v)), "irs_hip_batch_path")
        return int(v.value)

    def set_paired_tiles(self, enable=True):
        """Joined plain disjunctions on paired doc tiles: 1 / True = where it pays (default: segments
        of ~2.4 M docs and more), 2 = whatever the size, 0 / False = never (32-bit tiles).  The results
        are bit-identical either way."""
        _lib.check(self.L, self.L.irs_hip_batch_set_paired_tiles(self.handle, int(enable)),
                   "irs_hip_batch_set_paired_tiles")
        return self

    def paired_tiles(self):
        """Whether the last run's joined plain disjunctions ran on paired tiles."""
        v = C.c_int(0)
        _lib.check(self.L, self.L.irs_hip_batch_paired_tiles(self.handle, C.byref(v)),
                   "irs_hip_batch_paired_tiles")
        return bool(v.value)

    def set_wand(self, enable=True):
        """ExecutionContext::wand (index-search --search-mode wand): block-max pruning."""
        _lib.check(self.L, self.L.irs_hip_batch_set_wand(self.handle, int(enable)),
                   "irs_hip_batch_set_wand")
        return self

    def set_min_scores(self, min_scores):
        """irs::score::Min per query (the harness heap's k-th score so far); None clears."""
        if min_scores is None:
            ptr = None
        else:
            arr = np.ascontiguousarray(min_scores, np.float32)
            assert arr.shape == (self.nq_user,)
            ptr = arr.ctypes.data
        _lib.check(self.L, self.L.irs_hip_batch_set_min_scores(self.handle, ptr),
                   "irs_hip_batch_set_min_scores")
        return self

    def profile(self, enable=True):
        """True / 1: kernel timings; 2: count what the block-driven kernels decode; 3: both."""
        _lib.check(self.L, self.L.irs_hip_batch_profile(self.handle, int(enable)),
                   "irs_hip_batch_profile")
        return self

    def plan(self, stream=None):
        """Queue the planning stage of the next run on `stream` (irs_hip_batch_plan)."""
        _lib.check(self.L, self.L.irs_hip_batch_plan(self.handle, stream), "irs_hip_batch_plan")
        return self

    def run(self, stream=None):
        _lib.check(self.L, self.L.irs_hip_batch_run(self.handle, stream), "irs_hip_batch_run")
        return self

    def timings(self):
        ms = (C.c_float * _lib.K_COUNT)()
        _lib.check(self.L, self.L.irs_hip_batch_timings(self.handle, ms), "irs_hip_batch_timings")
        return [float(x) for x in ms]

    def reruns(self) -> int:
        n = C.c_uint32()
        _lib.check(self.L, self.L.irs_hip_batch_reruns(self.handle, C.byref(n)),
                   "irs_hip_batch_reruns")
        return n.value

    def touched(self):
        """(`.doc` + norm bytes decoded, positions read) by the last run (And / by_phrase)."""
        a, p = C.c_uint64(), C.c_uint64()
        _lib.check(self.L, self.L.irs_hip_batch_touched(self.handle, C.byref(a), C.byref(p)),
                   "irs_hip_batch_touched")
        return a.value, p.value

    def work(self):
        a, p = C.c_uint64(), C.c_uint64()
        _lib.check(self.L, self.L.irs_hip_batch_work(self.handle, C.byref(a), C.byref(p)),
                   "irs_hip_batch_work")
        return a.value, p.value

    def results(self):
        hits = np.zeros((self.nq, self.k), HIT)
        counts = np.zeros(self.nq, np.uint32)
        totals = np.zeros(self.nq, np.uint64)
        _lib.check(self.L, self.L.irs_hip_batch_results(
            self.handle, hits.ctypes.data, self.k, counts.ctypes.data, totals.ctypes.data),
            "irs_hip_batch_results")
        if self.multi:
            n = len(self.segs)
            return (hits.reshape(n, self.nq_user, self.k), counts.reshape(n, self.nq_user),
                    totals.reshape(n, self.nq_user))
        return hits, counts, totals

    def results_to_host(self, stream=None):
        """Queue the copy of the checked results to page-locked host memory
        (irs_hip_batch_results_to_host); host_results() waits for it."""
        _lib.check(self.L, self.L.irs_hip_batch_results_to_host(self.handle, stream),
                   "irs_hip_batch_results_to_host")
        return self

    def host_results(self):
        """(hits HIT[nq][k_max], counts[nq], totals[nq]) as numpy VIEWS of the batch's page-locked
        result memory: valid until the batch is closed, run or copied again."""
        hp, cp, tp, ks = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint32()
        _lib.check(self.L, self.L.irs_hip_batch_host_results(
            self.handle, C.byref(hp), C.byref(ks), C.byref(cp), C.byref(tp)),
            "irs_hip_batch_host_results")
        k = int(ks.value)
        hits = np.frombuffer((C.c_uint8 * (self.nq * k * HIT.itemsize)).from_address(hp.value), HIT)
        counts = np.frombuffer((C.c_uint8 * (self.nq * 4)).from_address(cp.value), np.uint32)
        totals = np.frombuffer((C.c_uint8 * (self.nq * 8)).from_address(tp.value), np.uint64)
        hits = hits.reshape(self.nq, k)
        if self.multi:
            n = len(self.segs)
            return (hits.reshape(n, self.nq_user, k), counts.reshape(n, self.nq_user),
                    totals.reshape(n, self.nq_user))
        return hits, counts, totals

    def device_results(self):
        dh, dc, km = C.c_void_p(), C.c_void_p(), C.c_uint32()
        _lib.check(self.L, self.L.irs_hip_batch_device_results(
            self.handle, C.byref(dh), C.byref(dc), C.byref(km)), "irs_hip_batch_device_results")
        return dh.value, dc.value, km.value

    def results_to_device(self, d_hits: int, d_counts: int, stream=None):
        _lib.check(self.L, self.L.irs_hip_batch_results_to_device(
            self.handle, d_hits, d_counts, stream), "irs_hip_batch_results_to_device")

    def close(self):
        if self.handle:
            self.L.irs_hip_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# -------------------------------------------------------------------- index --

@dataclass
class SegmentStats:
    """What by_term::prepare reads from a segment without touching postings."""
    docs_with_field: int
    total_term_freq: int
    docs_count: np.ndarray  # per term ordinal: term_meta::docs_count


def prepare(filters, scorer, segment_stats):
    """filter::prepare for a list of filters against ALL segments: statistics are
    index-global (term_filter.cpp:102-125): D = sum docs_with_field,
    d = sum docs_count of the term, avgdl from the summed field frequency."""
    dwf = sum(s.docs_with_field for s in segment_stats)
    ttf = sum(s.total_term_freq for s in segment_stats)
    out = []
    for flt in filters:
        if isinstance(flt, by_phrase):
            # FixedPrepareCollect (phrase_filter.cpp:212-293): term_stats.finish() of every
            # phrase term lands in ONE stats blob — BM25::collect / TFIDF::collect do
            # `idf +=` (bm25.cpp:381-383, tfidf.cpp:272-275), the norm constants are equal
            idf = f32(0)
            stats = None
            for t in flt.terms:
                dwt = sum(int(st.docs_count[t]) for st in segment_stats
                          if 0 <= t < len(st.docs_count))
                stats = scorer.collect(dwf, dwt, ttf)
                idf = f32(idf + stats.idf)
            one = scorer.term_scorer(TermStats(idf, stats.norm_const, stats.norm_length),
                                     flt.boost)
            out.append(PreparedQuery(OP_PHRASE, list(flt.terms), [one] * len(flt.terms), 0,
                                     [int(o) for o in flt.offsets]))
            continue
        op, subs = _terms_of(flt)
        scorers = []
        for s in subs:
            dwt = 0
            for st in segment_stats:
                if 0 <= s.term < len(st.docs_count):
                    dwt += int(st.docs_count[s.term])
            stats = scorer.collect(dwf, dwt, ttf)
            scorers.append(scorer.term_scorer(stats, s.boost))
        out.append(PreparedQuery(op, [s.term for s in subs], scorers,
                                 int(getattr(flt, "min_match", 0)),
                                 merge=int(getattr(flt, "merge", MERGE_SUM))))
    return out


def merge_topk_host(per_segment, k: int):
    """Global top-k over segments ordered (score desc, segment asc, doc asc) —
    the ordering tests/search/wand_test.cpp:72-86 fixes for the harness heap.
    per_segment: list of (hits HIT[nq][k], counts[nq]). Host-side variant used by
    tests; the GPU variant is irs_hip_merge_topk."""
    nq = len(per_segment[0][1])
    out = []
    for q in range(nq):
        rows = []
        for s, (hits, counts) in enumerate(per_segment):
            n = int(counts[q])
            for i in range(n):
                rows.append((-float(hits[q, i]["score"]), s, int(hits[q, i]["doc"])))
        rows.sort()
        out.append([(-a, s, d) for a, s, d in rows[:k]])
    return out


# ---------------------------------------------- scored multi-term expansion --
# by_prefix / by_wildcard / by_range WITH scorers (MultiTermQuery, multiterm_query.cpp:112-184):
# the filter's term visitor hands every visited term of every segment to a
# limited_sample_collector (limited_sample_collector.hpp:43-120) which keeps the
# `scored_terms_limit` (segment, term) states with the LONGEST postings — key = (docs_count, then
# the term's offset in the segment's visit), larger wins, a newcomer replaces the smallest only
# when strictly larger — as SCORED states; everything else is unscored.  execute() is then a
# disjunction (min_match 1, Sum) of one scored iterator per scored state and ONE
# lazy_bitset_iterator over the unscored terms: a doc matches when any visited term holds it, its
# score is the sum over the scored terms holding it (0 for docs only unscored terms hold).
# Statistics (limited_sample_collector::score :123-165): per distinct term the field statistics of
# the WHOLE index and term statistics collected from the segments where the term is SCORED only.
#
# On this path: the scored part is an ordinary Or of <= IRS_HIP_MAX_TERMS by_term filters — the
# scored_terms_limit of the reference's benchmark (scripts/search-benchmark.sh:
# --scored-terms-limit=16) fits — with a term absent from the segments where it is unscored; the
# unscored part and the total are irs_hip_bit_union, as in the reference.

def scored_states(visits_docs_counts, limit: int):
    """limited_sample_collector::collect over the segments' visits in order.
    visits_docs_counts[s] = docs_count of the terms segment s's visit yields, in visit order.
    -> sorted list of (segment, offset) that end up scored."""
    # (which of two EQUAL keys — the same docs_count at the same visit offset in two segments — is
    # the heap's root is decided by std::push_heap / std::pop_heap themselves: the binary heap below
    # moves its elements exactly as libstdc++'s does, so ties resolve as in the reference)
    rows = []          # (frequency, offset, segment) per scored state
    order = []         # binary min-heap of indices into rows
    above = lambda i, j: rows[j][:2] < rows[i][:2]            # row i sorts behind row j

    def sift_in(hole, idx):                                    # std::__push_heap
        while hole > 0 and above(order[(hole - 1) // 2], idx):
            order[hole] = order[(hole - 1) // 2]
            hole = (hole - 1) // 2
        order[hole] = idx

    def drop_root_to_back():                                   # std::pop_heap
        idx, order[-1] = order[-1], order[0]
        n, hole, child = len(order) - 1, 0, 0
        while child < (n - 1) // 2:
            child = 2 * (child + 1)
            if above(order[child], order[child - 1]):
                child -= 1
            order[hole] = order[child]
            hole = child
        if n % 2 == 0 and child == (n - 2) // 2:
            child = 2 * (child + 1)
            order[hole] = order[child - 1]
            hole = child - 1
        sift_in(hole, idx)

    for s, counts in enumerate(visits_docs_counts):
        for off, f in enumerate(counts):
            if limit <= 0:
                continue
            if len(rows) < limit:
                rows.append((int(f), off, s))
                order.append(len(rows) - 1)
                sift_in(len(order) - 1, order[-1])
            elif rows[order[0]][:2] < (int(f), off):         # strictly larger keys only
                root = order[0]
                drop_root_to_back()
                rows[root] = (int(f), off, s)
                sift_in(len(order) - 1, order[-1])
    return sorted((s, off) for _, off, s in rows)


def _scored_states_fast(visits_docs_counts, limit: int):
    """scored_states without the heap where the outcome does not depend on it: the collector keeps
    the `limit` largest keys (docs_count, offset); only when the key at the cut also occurs right
    behind it (the same docs_count at the same offset in two segments) the heap's own order
    decides, and the emulation runs."""
    if limit <= 0:
        return []
    seg = np.concatenate([np.full(len(c), s, np.int64) for s, c in enumerate(visits_docs_counts)] or [np.zeros(0, np.int64)])
    off = np.concatenate([np.arange(len(c), dtype=np.int64) for c in visits_docs_counts] or [np.zeros(0, np.int64)])
    frq = np.concatenate([np.asarray(c, np.int64) for c in visits_docs_counts] or [np.zeros(0, np.int64)])
    if frq.size <= limit:
        return sorted(zip(seg.tolist(), off.tolist()))
    order = np.lexsort((off, frq))[::-1]          # by (docs_count, offset), largest first
    a, b = order[limit - 1], order[limit]
    if frq[a] == frq[b] and off[a] == off[b]:
        return scored_states(visits_docs_counts, limit)
    top = order[:limit]
    return sorted(zip(seg[top].tolist(), off[top].tolist()))


class PreparedExpansion:
    """One scored multi-term filter prepared against all segments."""

    def __init__(self, scored, scored_in, unscored_in, scorers, visited_in=None, slots=None,
                 present=None, c0=None):
        self.scored = scored            # distinct scored term ordinals (query term slots)
        self.scored_in = scored_in      # [segment] -> set of scored ordinals there
        self._unscored_in = unscored_in  # (None: worked out from visited_in when asked for)
        self.scorers = scorers          # (kind, c0, norm_const, norm_length) per slot
        self.visited_in = visited_in    # [segment] -> np.uint32 ordinals the visitor yielded there
        self.slots = slots              # `scored` as np.uint32
        self.present = present          # bool [segment][slot]: the slot's term is scored in the segment
        self.c0 = c0                    # float32 per slot (scorers[j][1])

    @property
    def unscored_in(self):
        """[segment] -> np.uint32 ordinals that are visited but unscored there (visit order)."""
        if self._unscored_in is None:
            self._unscored_in = [
                va[~np.isin(va, np.fromiter(sc, np.uint32, len(sc)))] if sc else va
                for va, sc in zip(self.visited_in, self.scored_in)]
        return self._unscored_in

    def n_unscored(self, s: int) -> int:
        return len(self.visited_in[s]) - len(self.scored_in[s]) if self._unscored_in is None \
            else len(self._unscored_in[s])


def _top_offsets_padded(visit_arrays, lens, flat_counts, limit):
    """-> (rows, cols) of the `limit` largest (docs_count, offset) keys of every visit, offsets
    ascending within a row; flat_counts = the docs_counts of the concatenated visits."""
    nq = len(visit_arrays)
    width = int(lens.max()) if nq else 0
    if limit <= 0 or width == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    inside = np.arange(width)[None, :] < lens[:, None]       # row-major order = concatenation order
    key = np.full((nq, width), -1, np.int64)
    key[inside] = flat_counts * (1 << 24) + np.broadcast_to(np.arange(width), (nq, width))[inside]
    if width > limit:
        top = np.argpartition(key, width - limit, axis=1)[:, width - limit:]
    else:
        top = np.broadcast_to(np.arange(width), (nq, width)).copy()
    top.sort(axis=1)                                          # offsets ascending = ordinals ascending
    ok = np.take_along_axis(key, top, axis=1) >= 0
    rows = np.broadcast_to(np.arange(nq)[:, None], top.shape)[ok]
    return rows, top[ok]


def _top_offsets_one_segment(visit_arrays, counts_of, limit):
    """The collector's choice for MANY filters over ONE segment at once: offsets are unique within
    a visit, so it keeps the `limit` largest (docs_count, offset) keys — no heap order involved.
    -> (rows, cols, lens): for filter q the scored visit offsets cols[rows == q], ascending.
    Long visits (a wildcard: thousands of mostly rare terms) first drop everything below a
    docs_count that still leaves every filter its `limit` largest: the keys are then sorted out
    among a few percent of the elements."""
    nq = len(visit_arrays)
    lens = np.fromiter((len(v) for v in visit_arrays), np.int64, nq)
    if limit <= 0 or nq == 0 or int(lens.max()) == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), lens
    flat = np.concatenate(visit_arrays)
    cnt = counts_of[flat]
    keep_about = 4 * limit * nq
    if len(flat) <= 4 * keep_about:
        rows, cols = _top_offsets_padded(visit_arrays, lens, cnt, limit)
        return rows, cols, lens
    # a docs_count that about 4 * limit elements per filter reach (from a sample)
    sample = cnt[:: max(1, len(flat) // 16384)]
    cut = np.partition(sample, len(sample) - 1 - min(len(sample) - 1, (keep_about * len(sample)) // len(flat)))[
        len(sample) - 1 - min(len(sample) - 1, (keep_about * len(sample)) // len(flat))]
    at = np.flatnonzero(cnt >= cut)
    ends = np.cumsum(lens)
    row_at = np.searchsorted(ends, at, side="right")
    above = np.bincount(row_at, minlength=nq)
    settled = (above >= limit) | (above == lens)       # the `limit` largest are all at or above the cut
    rows_out, cols_out = [], []
    if settled.any():
        sel = settled[row_at]
        sub_rows_all, sub_at = row_at[sel], at[sel]
        ids = np.flatnonzero(settled)
        # the kept elements of the settled filters as (shorter) visits of their own: positions in
        # the original visit are what the keys need, so they ride along
        sub_lens = above[ids]
        sub_col = sub_at - (ends - lens)[sub_rows_all]            # offset in the original visit
        width = int(sub_lens.max())
        inside = np.arange(width)[None, :] < sub_lens[:, None]
        key = np.full((len(ids), width), -1, np.int64)
        key[inside] = cnt[sub_at] * (1 << 24) + sub_col
        if width > limit:
            top = np.argpartition(key, width - limit, axis=1)[:, width - limit:]
        else:
            top = np.broadcast_to(np.arange(width), (len(ids), width)).copy()
        picked = np.take_along_axis(key, top, axis=1)
        picked = np.where(picked >= 0, picked & ((1 << 24) - 1), 1 << 30)     # -> original offsets
        picked.sort(axis=1)
        ok = picked < (1 << 30)
        rows_out.append(np.broadcast_to(ids[:, None], picked.shape)[ok])
        cols_out.append(picked[ok])
    rest = np.flatnonzero(~settled)
    if rest.size:                                       # (few: the cut was too high for them)
        starts = ends - lens
        sub = [visit_arrays[q] for q in rest]
        sub_cnt = np.concatenate([cnt[starts[q]:ends[q]] for q in rest])
        r, c = _top_offsets_padded(sub, lens[rest], sub_cnt, limit)
        rows_out.append(rest[r])
        cols_out.append(c)
    rows, cols = np.concatenate(rows_out), np.concatenate(cols_out)
    order = np.lexsort((cols, rows))
    return rows[order], cols[order], lens


def prepare_expansions(visits, limit, scorer, segment_stats, boost=1.0):
    """filter::prepare of scored multi-term filters: visits[q][s] = term ordinals the filter's
    visitor yields in segment s (ascending term order, as term_reader::iterator() enumerates).
    The statistics of all scored terms of all filters are worked out at once (numpy), value for
    value what scorer.collect / term_scorer yield one term at a time."""
    dwf = sum(s.docs_with_field for s in segment_stats)
    ttf = sum(s.total_term_freq for s in segment_stats)
    dcs = [np.asarray(st.docs_count) for st in segment_stats]
    n_segs = len(segment_stats)
    parts, all_dwt = [], []
    if n_segs == 1 and len(visits):
        # one segment: every filter's choice in a few array operations
        vas = [np.asarray(per_seg[0], np.uint32) for per_seg in visits]
        rows, cols, lens = _top_offsets_one_segment(vas, dcs[0].astype(np.int64), limit)
        starts = np.cumsum(lens) - lens
        flat = np.concatenate(vas) if len(vas) else np.zeros(0, np.uint32)
        at = starts[rows] + cols
        scored_flat = flat[at]
        all_dwt = dcs[0][scored_flat.astype(np.int64)].astype(np.float64)
        per_q = np.bincount(rows, minlength=len(vas)) if len(rows) else np.zeros(len(vas), np.int64)
        scored_parts = np.split(scored_flat, np.cumsum(per_q)[:-1])
        for va, sp in zip(vas, scored_parts):
            parts.append((sp, None, None, [va]))
    else:
        for per_seg in visits:
            counts = [dcs[s][np.asarray(v, np.int64)] if len(v) else np.zeros(0, np.int64)
                      for s, v in enumerate(per_seg)]
            states = _scored_states_fast(counts, limit)
            scored_in = [set() for _ in per_seg]
            for s, off in states:
                scored_in[s].add(int(per_seg[s][off]))
            slots = sorted(set().union(*scored_in)) if scored_in else []
            # term statistics from the segments where the term is scored (collector::score)
            all_dwt.extend(sum(int(dcs[s][t]) for s in range(len(per_seg)) if t in scored_in[s]) for t in slots)
            unscored_in, visited_in = [], []
            for s, v in enumerate(per_seg):
                va = np.asarray(v, np.uint32)
                keep = ~np.isin(va, np.fromiter(scored_in[s], np.uint32, len(scored_in[s]))) if scored_in[s] else \
                    np.ones(len(va), bool)
                unscored_in.append(va[keep])
                visited_in.append(va)
            parts.append((np.asarray(slots, np.uint32), scored_in, unscored_in, visited_in))
    dwt = np.asarray(all_dwt, np.float64)
    if isinstance(scorer, BM25):
        idf = np.log1p((float(dwf) - dwt + 0.5) / (dwt + 0.5)).astype(np.float32)
        probe = scorer.collect(dwf, 1, ttf)
        kind = scorer.term_scorer(probe)[0]
        nc, nl = probe.norm_const, probe.norm_length
        c0 = (f32(f32(boost) * f32(scorer.k + f32(1))) * idf).astype(np.float32)
    else:
        idf = np.log1p((dwf + 1.0) / (dwt + 1.0)).astype(np.float32)
        kind, nc, nl = scorer.term_scorer(TermStats(f32(1)))[0], f32(0), f32(0)
        c0 = (f32(boost) * idf).astype(np.float32)
    out, at = [], 0
    for slots, scored_in, unscored_in, visited_in in parts:
        n = len(slots)
        mine = c0[at:at + n]
        at += n
        slot_list = slots.tolist()
        if scored_in is None:                       # (one segment: every slot is scored there)
            scored_in = [set(slot_list)]
            present = np.ones((1, n), bool)
        else:
            present = np.array([[t in sc for t in slot_list] for sc in scored_in], bool).reshape(n_segs, n)
        out.append(PreparedExpansion(slot_list, scored_in, unscored_in,
                                     [(kind, x, nc, nl) for x in mine], visited_in, slots, present, mine))
    return out


def expansion_arrays(segs, prepared, k):
    """The scored parts of a list of prepared expansions as ONE batch: an Or per filter whose term
    slots are absent (NO_TERM) in the segments where the term is unscored.  Filters without any
    scored term get a one-slot query of an absent term (matches nothing)."""
    n_segs, nq = len(segs), len(prepared)
    n_slots = np.fromiter((max(1, len(p.scored)) for p in prepared), np.int64, nq)
    first = np.cumsum(n_slots) - n_slots
    n_entries = int(n_slots.sum())
    queries = np.zeros(nq, QUERY)
    queries["op"], queries["n_terms"], queries["first_term"] = OP_OR, n_slots, first
    queries["k"], queries["min_match"], queries["merge"] = int(k), 1, MERGE_SUM
    terms = np.zeros((n_segs, n_entries), TERM_SCORER)
    terms["term"] = NO_TERM                      # (also the one slot of a filter without scored terms)
    terms["kind"] = SCORE_BM1
    real = [q for q, p in enumerate(prepared) if p.scored]
    if real:
        where = np.concatenate([first[q] + np.arange(len(prepared[q].scored)) for q in real])
        slots = np.concatenate([prepared[q].slots for q in real])
        present = np.concatenate([prepared[q].present for q in real], axis=1)      # [seg][slot]
        kind, _, nc, nl = prepared[real[0]].scorers[0]
        terms["term"][:, where] = np.where(present, slots[None, :], np.uint32(NO_TERM))
        terms["kind"][:, where] = kind
        terms["c0"][:, where] = np.concatenate([prepared[q].c0 for q in real])[None, :]
        terms["norm_const"][:, where] = nc
        terms["norm_length"][:, where] = nl
    return QueryArrays(n_segs, queries, terms, k)


def execute_expansions(readers, prepared, k):
    """MultiTermQuery::execute for every (filter, segment) + the harness's top k per segment:
    hits HIT[n_segs][nq][k], counts, totals.  The scored disjunctions are one batch; a segment's
    total is the population of the union of ALL visited terms' postings (bit_union, like
    lazy_bitset_iterator + the scored iterators); where fewer than k docs score, the list is filled
    with the docs only unscored terms hold (score 0, ascending doc id: this path's tie order)."""
    n_segs, nq = len(readers), len(prepared)
    b = QueryBatch(readers, expansion_arrays(readers, prepared, k))
    hits, counts, totals = (x.copy() for x in b.run().results())
    b.close()
    hits = hits.reshape(n_segs, nq, k)
    counts = counts.reshape(n_segs, nq)
    totals = totals.reshape(n_segs, nq)
    for s, sr in enumerate(readers):
        n_words = (sr.num_docs + 64) // 64
        # the totals: the population of the union of ALL visited terms, for every filter that has
        # unscored terms here, in one call (only the counts cross PCIe)
        need = [q for q, p in enumerate(prepared) if p.n_unscored(s)]
        visited = [prepared[q].visited_in[s] for q in need]
        if need:
            totals[s, need] = sr.bit_union_counts(visited)
        for q, v in zip(need, visited):
            have = int(counts[s, q])
            if have >= k or int(totals[s, q]) <= have:
                continue
            # fewer than k docs score: the rest of the list are docs only unscored terms hold
            p = prepared[q]
            only, _ = sr.bit_union(v, n_words)
            if p.scored_in[s]:
                sc, _ = sr.bit_union(np.array(sorted(p.scored_in[s]), np.uint32), n_words)
                only = only & ~sc
            docs = np.nonzero(np.unpackbits(only.view(np.uint8), bitorder="little"))[0][:k - have]
            hits[s, q, have:have + len(docs)]["doc"] = docs
            hits[s, q, have:have + len(docs)]["score"] = 0.0
            counts[s, q] = have + len(docs)
    return hits, counts, totals
