"""Test bodies shared by the emulator tier (CPU, not gpu) and the GPU tier:
each takes the bound C-ABI library `L` and a size scale."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import pytest

import oracle
import parity
from iresearch_amd import _lib, search, synth
from iresearch_amd.search import BM25, TFIDF, And, Or, by_phrase, by_term

GOLDEN = __import__("pathlib").Path(__file__).parent / "golden"


def open_lists(L, lists, num_docs, layout, norms=None):
    seg = synth.segment_from_lists(lists, num_docs, layout, norms)
    return seg, search.SegmentReader.from_synth(seg, L=L)


def assert_decode(sr, seg, term, docs, freqs):
    d, f = sr.decode_term(term)
    assert np.array_equal(d, np.asarray(docs, np.uint32)), ("docs", term)
    assert np.array_equal(f, np.asarray(freqs, np.uint32)), ("freqs", term)
    od, of = oracle.decode_term(seg.doc_file, seg.metas[term], seg.layout)
    assert np.array_equal(d, od) and np.array_equal(f, of), ("oracle", term)
    d2, f2 = sr.decode_term(term, want_freq=False)
    assert np.array_equal(d2, d) and f2 is None


def case_wand_equals_exhaustive(L, num_docs=60_000, max_rank=256, layout=synth.LAYOUT_SIMD4,
                                ks=(10, 100)):
    """The reference's differential rule for WAND (tests/search/wand_test.cpp:231-241): the
    top k of a run that may skip blocks is the top k of the exhaustive run — same docs, same
    scores, same order — for term / And / Or queries under BM25, BM15, BM11 and TF-IDF, on
    the default corpus and on a clustered one (bursty posting lists, where whole blocks and
    tiles really are skipped).  Total hits may only shrink."""
    scorers = [BM25(), BM25(1.2, 0.0), BM25(1.2, 1.0), TFIDF(False), TFIDF(True)]
    pruned = 0
    for clustered, wand_count in ((False, 0), (True, 0), (True, 1)):
        # (wand_count 1: the field was indexed with a scorer — the bounds then come from the
        # index's own wand data, k_wand_skip0, instead of being derived from the postings)
        kw = dict(topic_docs=2048, topic_percent=85, topic_terms=12) if clustered else {}
        seg = synth.build_segment(num_docs, max_rank, layout=layout, wand_count=wand_count, **kw)
        sr = search.SegmentReader.from_synth(seg, L=L)
        assert (sr.wand_source()[0] > 0) == (wand_count > 0)
        stats = [parity.segment_stats(seg)]
        rng = np.random.default_rng(11 + clustered)
        filters = [by_term(int(t)) for t in rng.integers(0, max_rank, 4)]
        filters += [And([by_term(int(t)) for t in rng.choice(max_rank // 4, 2, replace=False)])
                    for _ in range(6)]
        filters += [And([by_term(int(t)) for t in rng.choice(max_rank // 2, 3, replace=False)])
                    for _ in range(4)]
        filters += [Or([by_term(int(t)) for t in rng.choice(max_rank, 3, replace=False)])
                    for _ in range(4)]
        for scorer in scorers:
            prep = search.prepare(filters, scorer, stats)
            for k in ks:
                # like with like, bit for bit: the work-item / block-driven kernels (where the
                # pruning happens), and whatever the cost rules deal the units to (a unit on
                # joined streams runs exhaustively with or without the option)
                for path in (_lib.PATH_ITEMS, _lib.PATH_AUTO):
                    ex = sr.batch(prep, k).set_path(path)
                    h0, c0, t0 = ex.run().results()
                    ex.close()
                    wb = sr.batch(prep, k).set_path(path).set_wand(True)
                    h1, c1, t1 = wb.run().results()
                    wb.close()
                    assert np.array_equal(c0, c1), (clustered, type(scorer).__name__, k)
                    for q in range(len(filters)):
                        n = int(c0[q])
                        assert np.array_equal(h0[q, :n], h1[q, :n]), (clustered, q, k, path)
                    assert (t1 <= t0).all()
                    pruned += int((t0 - t1).sum())
            # and the exhaustive run is the oracle's
            parity.check_single_segment(seg, filters, scorer, ks[-1], h0, c0, t0)
        sr.close()
    assert pruned > 0, "no block or tile was ever skipped: the pruning path went untested"


# ------------------------------------------------------------------ decode --

def case_decode_reference_lists(L, layout):
    """The literal lists of tests/formats/formats_10_tests.cpp:452-457 and the
    6098 doc ids of tests/resources/postings.txt (ires336, :775-864)."""
    docs0 = [1, 3, 5, 7, 79, 101, 124]
    docs1 = [2, 7, 9, 19]
    p6098 = np.loadtxt(GOLDEN / "postings_6098.txt", dtype=np.uint32)
    assert p6098.size == 6098
    lists = [(docs0, [10] * 7), (docs1, [10] * 4), (p6098, np.ones(6098, np.uint32)),
             (p6098, (np.arange(6098) % 7 + 1).astype(np.uint32))]
    seg, sr = open_lists(L, lists, int(p6098.max()) + 10, layout)
    for t, (d, f) in enumerate(lists):
        assert_decode(sr, seg, t, d, f)
    # block directory == skip level 0 written by the emitter (formats_10.cpp:501-533)
    last, offs = sr.term_directory(2)
    sl, sp, levels = oracle.read_skip0(seg.doc_file, seg.metas[2])
    assert len(last) == 6098 // 128 == 47 and levels >= 1
    assert np.array_equal(last[:len(sl)], sl) and np.array_equal(offs[1:], sp[:len(offs) - 1])
    assert np.array_equal(last, p6098[127::128][:47])
    sr.close()


def case_decode_sizes(L, layout, sizes=(1, 2, 117, 127, 128, 129, 255, 256, 319, 1024, 10_000)):
    """Sizes of tests/formats/formats_15_tests.cpp:695-764 plus block edges; freqs
    ~ Normal(50, 14) as GenerateDocs (:553-567)."""
    rng = np.random.default_rng(7)
    lists = []
    n_docs = 200_000
    for n in sizes:
        docs = np.sort(rng.choice(np.arange(1, n_docs + 1), n, replace=False)).astype(np.uint32)
        freqs = np.clip(np.rint(rng.normal(50, 14, n)), 1, None).astype(np.uint32)
        lists.append((docs, freqs))
    seg, sr = open_lists(L, lists, n_docs, layout)
    for t, (d, f) in enumerate(lists):
        assert_decode(sr, seg, t, d, f)
    sr.close()


def case_decode_edge_blocks(L, layout):
    """All-equal doc and freq blocks (bitpack.hpp:82-86), every bit width up to the
    maxima (31-bit gaps, 32-bit freqs), freq == 1 tails, first doc == 1."""
    lists = []
    n_docs = 0x7FFE0000
    # consecutive docs from 1: all deltas equal (first delta = doc - 1 = 0 breaks it: start at 2)
    d = np.arange(2, 2 + 300, dtype=np.uint32)
    lists.append((d, np.full(300, 3, np.uint32)))             # all-equal deltas & freqs
    d = np.arange(1, 1 + 256, dtype=np.uint32)
    lists.append((d, np.ones(256, np.uint32)))                  # doc 1 first: delta 0 then 1s
    for bits in (1, 2, 5, 8, 13, 17, 24, 29, 31):
        rng = np.random.default_rng(bits)
        gaps = rng.integers(1, min(1 << bits, 1 << 22), 200, dtype=np.int64)
        gaps[0] = min((1 << bits) - 1, 1 << 22)
        d = np.cumsum(gaps).astype(np.uint32)
        fb = min(bits + 1, 32)
        f = rng.integers(1, (1 << fb) - 1, 200, dtype=np.uint64).astype(np.uint32)
        f[5] = (1 << fb) - 1
        lists.append((d, f))
    # one huge gap: 31-bit delta
    d = np.concatenate([np.arange(1, 129), [0x7FFD0000], 0x7FFD0000 + np.arange(1, 140)]).astype(np.uint32)
    f = np.ones(d.size, np.uint32)
    f[::3] = 0xFFFFFFFF
    lists.append((d, f))
    seg, sr = open_lists(L, lists, n_docs, layout, norms=False)
    for t, (dd, ff) in enumerate(lists):
        assert_decode(sr, seg, t, dd, ff)
    sr.close()


def case_join_edge_blocks(L, layout):
    """The joined-stream path (k_join: every list decoded once per batch) over awkward lists:
    all-equal doc and freq blocks, every doc bit width a 200 k-doc segment allows, exactly 128 /
    129 / 256 postings, tail-only and single-doc lists, lists that start late, end early or
    jump over several 12288-doc tiles, frequencies of long documents (64 .. 255: the entry's two
    low bits; 256 does not fit an entry: that query runs as work items whatever the path asked for)
    — each list alone and all of them in one Or, against the oracle and bit for bit against the
    work-item path."""
    n_docs = 200_000
    rng = np.random.default_rng(77)
    lists = []
    lists.append((np.arange(2, 2 + 300, dtype=np.uint32), np.full(300, 3, np.uint32)))   # all-equal
    lists.append((np.arange(1, 1 + 256, dtype=np.uint32), np.ones(256, np.uint32)))
    for bits in (1, 2, 3, 5, 8, 10):
        n = 128 * 3 + bits
        gaps = rng.integers(1, 1 << bits, n, dtype=np.int64)
        gaps[7] = (1 << bits) - 1 if bits > 1 else 1
        d = (50 + np.cumsum(gaps)).astype(np.uint32)
        lists.append((d, rng.integers(1, 1 << min(bits, 6), n).astype(np.uint32)))
    for n in (1, 5, 127, 128, 129, 256, 257):
        d = np.sort(rng.choice(np.arange(1, n_docs + 1), n, replace=False)).astype(np.uint32)
        lists.append((d, rng.integers(1, 64, n).astype(np.uint32)))      # up to 63: general form
    late = np.sort(rng.choice(np.arange(5 * 12288 + 100, 7 * 12288), 300, replace=False))
    lists.append((late.astype(np.uint32), rng.integers(1, 4, 300).astype(np.uint32)))   # starts late, ends early
    jump = np.concatenate([np.arange(10, 138), [3 * 12288 + 5], 9 * 12288 + np.arange(1, 200), [n_docs]])
    lists.append((jump.astype(np.uint32), np.ones(jump.size, np.uint32)))               # jumps over tiles
    lists.append((np.array([n_docs], np.uint32), np.array([63], np.uint32)))            # last doc only
    lists.append((np.array([12288], np.uint32), np.array([2], np.uint32)))              # a tile's last doc
    lists.append((np.array([12289], np.uint32), np.array([2], np.uint32)))              # a tile's first doc
    d = np.sort(rng.choice(np.arange(1, n_docs + 1), 700, replace=False)).astype(np.uint32)
    f = rng.integers(1, 256, 700).astype(np.uint32)
    f[:6] = (63, 64, 65, 127, 128, 255)
    lists.append((d, f))                                                                # long documents
    i_wide = len(lists) - 1
    lists.append((d[::3].copy(), np.full(d[::3].size, 256, np.uint32)))                 # ... too long for an entry
    norms = rng.integers(1, 256, n_docs).astype(np.uint8)
    seg, sr = open_lists(L, lists, n_docs, layout, norms)
    nt = len(lists)
    filters = [by_term(t) for t in range(nt)]
    filters += [Or([by_term(t) for t in range(0, nt, 2)][:16]), Or([by_term(t) for t in range(1, nt, 2)][:16]),
                Or([by_term(8), by_term(nt + 3), by_term(12)]),
                Or([by_term(i_wide), by_term(3), by_term(10)]), Or([by_term(i_wide + 1), by_term(i_wide)])]
    for scorer in (BM25(), TFIDF(True), BM25(1.2, 0.0)):
        # the long documents' list runs on joined streams when it can choose
        prep = search.prepare([by_term(i_wide), Or([by_term(i_wide), by_term(3)])], scorer,
                              [parity.segment_stats(seg)])
        b = sr.batch(prep, 10).run()
        assert b.path() == _lib.PATH_JOINED
        b.close()
        for k in (3, 500):
            hj = run_and_check(L, seg, filters, scorer, k, sr=sr, path=_lib.PATH_JOINED)
            hi = run_and_check(L, seg, filters, scorer, k, sr=sr, path=_lib.PATH_ITEMS)
            for x, y in zip(hj, hi):
                assert np.array_equal(x, y)
    sr.close()


def _doc_file(version, body: bytes) -> np.ndarray:
    """Header + postings + footer of a `.doc` file (format_utils.cpp:57-67; big-endian ints)."""
    name = b"iresearch_10_postings_documents"
    hdr = (0x3fd76c17).to_bytes(4, "big") + bytes([len(name)]) + name + version.to_bytes(4, "big")
    ftr = ((-0x3fd76c17) & 0xFFFFFFFF).to_bytes(4, "big") + (0).to_bytes(4, "big") + (0).to_bytes(8, "big")
    return np.frombuffer(hdr + body + ftr, np.uint8).copy(), len(hdr)


def case_decode_reference_packed(L, layout):
    """`.doc` blocks assembled BY HAND from words the reference's own compiled packers
    produced (tests/golden/codec_golden.npz: packed::pack_block / simdpackwithoutmask on the
    literal vector of tests/utils/bit_packing_tests.cpp:102-114 and on random data, bits
    1..32) and decoded through the C ABI: no emitter and no oracle in the loop — the
    expectation is the packers' INPUT."""
    from iresearch_amd._lib import TERM_META
    g = np.load(GOLDEN / "codec_golden.npz")
    lay = "simd4" if layout == synth.LAYOUT_SIMD4 else "scalar"

    def values(kind, bits):
        if kind == "lit":
            mask = np.uint32(0xFFFFFFFF if bits == 32 else (1 << bits) - 1)
            return g["literal"] & mask
        return g["random_b%d" % bits]

    def block(kind, bits):
        return bytes([bits]) + g["%s_%s_b%d" % (lay, kind, bits)].astype("<u4").tobytes()

    body = b""
    terms = []   # (doc_start relative to the body, [(doc kind, dbits, freq kind, fbits), ...])
    for kind in ("lit", "rnd"):
        for b in range(1, 33):
            # every width as a freq block; doc deltas up to 23 bits (128 of them stay < 2^31)
            terms.append((len(body), [(kind, min(b, 23), kind, b)]))
            body += block(kind, min(b, 23)) + block(kind, b)
    # a list of 8 blocks: the base carries over from block to block
    blocks = [("rnd" if i & 1 else "lit", 9 + i, "lit" if i & 1 else "rnd", 1 + 4 * i) for i in range(8)]
    terms.append((len(body), blocks))
    for dk, db, fk, fb in blocks:
        body += block(dk, db) + block(fk, fb)
    doc_file, hdr = _doc_file(5 if layout == synth.LAYOUT_SIMD4 else 4, body)
    metas = np.zeros(len(terms), TERM_META)
    want = []
    for t, (off, blks) in enumerate(terms):
        docs, freqs, base = [], [], 1   # doc_limits::min(), formats_10.cpp:636
        for dk, db, fk, fb in blks:
            d = base + np.cumsum(values(dk, db).astype(np.uint64))
            base = int(d[-1])
            docs.append(d)
            freqs.append(values(fk, fb))
        docs = np.concatenate(docs)
        assert docs[-1] < 2**31
        want.append((docs.astype(np.uint32), np.concatenate(freqs)))
        metas[t]["docs_count"] = docs.size
        metas[t]["freq"] = min(int(np.concatenate(freqs).astype(np.uint64).sum()), 0xFFFFFFFF)
        metas[t]["doc_start"] = hdr + off
        metas[t]["e_skip_start"] = 0   # the block directory never reads the skip data
    num_docs = max(int(w[0][-1]) for w in want)
    sr = search.SegmentReader(doc_file, metas, num_docs, layout, L=L)
    for t, (d, f) in enumerate(want):
        got_d, got_f = sr.decode_term(t)
        assert np.array_equal(got_d, d), ("docs", t)
        assert np.array_equal(got_f, f), ("freqs", t)
        last, offs = sr.term_directory(t)
        assert np.array_equal(last, d[127::128]), ("directory", t)
    sr.close()


def case_decode_synth(L, layout, num_docs, max_rank, step=1):
    seg = synth.build_segment(num_docs, max_rank, layout=layout, keep_postings=True)
    sr = search.SegmentReader.from_synth(seg, L=L)
    for r in range(1, max_rank + 1, step):
        d, f = seg.postings[r]
        got_d, got_f = sr.decode_term(r - 1)
        assert np.array_equal(got_d, d) and np.array_equal(got_f, f), r
        last, offs = sr.term_directory(r - 1)
        assert np.array_equal(last, d[127::128][:len(d) // 128]), r
    sr.close()
    return seg


# ----------------------------------------------------------------- queries --

def run_and_check(L, seg, filters, scorer, k, tile=0, stride=0, cap=0, sr=None, path=None):
    own = sr is None
    sr = sr or search.SegmentReader.from_synth(seg, L=L)
    prep = search.prepare(filters, scorer, [parity.segment_stats(seg)])
    b = sr.batch(prep, k)
    if tile or stride or cap:
        b.configure(tile, stride, cap)
    if path is not None:
        b.set_path(path)
    hits, counts, totals = b.run().results()
    parity.check_single_segment(seg, filters, scorer, k, hits, counts, totals)
    b.close()
    if own:
        sr.close()
    return hits, counts, totals


def standard_filters(max_rank, n_or8=4, seed=synth.SEED + 2):
    lo = max(2, max_rank // 256)
    ranks = synth.make_queries(n_or8, 8, lo, max_rank, seed)
    fl = [Or([by_term(int(r) - 1) for r in row]) for row in ranks]
    pairs = synth.make_queries(3, 2, lo, max_rank, seed + 1)
    fl += [Or([by_term(int(r) - 1) for r in row]) for row in pairs]
    fl += [by_term(max_rank // 2), by_term(0)]
    fl += [And([by_term(3), by_term(max_rank // 8), by_term(1)]),
           And([by_term(max_rank - 1), by_term(max_rank - 2)])]
    fl += [Or([by_term(max_rank - 1, 2.5), by_term(max_rank - 2), by_term(max_rank - 3, 0.5)])]
    # Or::min_match_count (MinMatchQuery, boolean_query.cpp:212-247)
    mm = [by_term(2), by_term(max_rank // 16), by_term(max_rank // 4), by_term(5), by_term(9)]
    fl += [Or(mm, min_match=2), Or(mm, min_match=4), Or(mm[:3], min_match=3),
           Or(mm[:2], min_match=3), Or(mm + [by_term(10 * max_rank)], min_match=5)]
    return fl


def case_queries_all_scorers(L, num_docs, max_rank, layout=synth.LAYOUT_SIMD4, ks=(10, 1000)):
    seg = synth.build_segment(num_docs, max_rank, layout=layout)
    sr = search.SegmentReader.from_synth(seg, L=L)
    filters = standard_filters(max_rank)
    for scorer in (BM25(), BM25(1.2, 0.0), BM25(2.0, 1.0), TFIDF(False), TFIDF(True)):
        for k in ks:
            run_and_check(L, seg, filters, scorer, k, sr=sr)
    sr.close()


def case_queries_tiles_and_strides(L, num_docs, max_rank):
    seg = synth.build_segment(num_docs, max_rank)
    sr = search.SegmentReader.from_synth(seg, L=L)
    filters = standard_filters(max_rank, n_or8=2)
    ref = None
    for tile in (4096, 6144, 8192, 12288):
        for stride in (1, 3, 16, 1000):
            hits, counts, totals = run_and_check(L, seg, filters, BM25(), 100, tile, stride, sr=sr)
            # results do not depend on the tiling or on the pilot sample
            if ref is None:
                ref = (hits.copy(), counts.copy(), totals.copy())
            else:
                assert np.array_equal(ref[1], counts) and np.array_equal(ref[2], totals)
                assert np.array_equal(ref[0], hits)
    sr.close()


def case_many_items(L, num_docs=40_000):
    """16 of the densest terms in one query: far more (term, block) work items per doc tile
    than the kernels stage at once (256), so the multi-round item path runs; also AND and
    MinMatch over the same terms, every tile size."""
    seg = synth.build_segment(num_docs, 32)
    sr = search.SegmentReader.from_synth(seg, L=L)
    terms = [by_term(t) for t in range(16)]
    filters = [Or(terms), And(terms[:6]), Or(terms, min_match=12), Or(terms[:9])]
    ref = None
    for tile in (4096, 6144, 8192, 12288):
        hits, counts, totals = run_and_check(L, seg, filters, BM25(), 200, tile, 3, sr=sr)
        if ref is None:
            ref = (hits.copy(), counts.copy(), totals.copy())
        else:
            assert np.array_equal(ref[0], hits) and np.array_equal(ref[1], counts)
            assert np.array_equal(ref[2], totals)
    run_and_check(L, seg, filters[:2], TFIDF(True), 50, sr=sr)
    sr.close()


def case_queries_ragged(L, layout=synth.LAYOUT_SIMD4):
    """Absent terms, single-doc terms, df < 128, df == 128, tail-only and
    block-only lists, k larger than the number of hits, empty results."""
    rng = np.random.default_rng(3)
    n_docs = 30_000
    def mk(n):
        d = np.sort(rng.choice(np.arange(1, n_docs + 1), n, replace=False)).astype(np.uint32)
        return d, rng.integers(1, 9, n).astype(np.uint32)
    lists = [mk(1), mk(5), mk(127), mk(128), mk(129), mk(256), mk(3000), mk(12_000), mk(1)]
    lists.append((np.array([n_docs], np.uint32), np.array([4], np.uint32)))      # last doc only
    lists.append((np.array([1], np.uint32), np.array([1], np.uint32)))           # first doc only
    norms = rng.integers(1, 256, n_docs).astype(np.uint8)
    seg, sr = open_lists(L, lists, n_docs, layout, norms)
    nt = len(lists)
    filters = [by_term(t) for t in range(nt)]
    filters += [Or([by_term(0), by_term(8)]), Or([by_term(t) for t in range(nt)]),
                Or([by_term(1), by_term(nt + 5), by_term(2)]),         # absent middle term
                Or([by_term(nt + 1), by_term(nt + 2)]),                # all absent: empty
                And([by_term(6), by_term(7)]), And([by_term(6), by_term(nt + 1)]),
                And([by_term(0), by_term(8)]), And([by_term(7), by_term(6), by_term(5)]),
                Or([by_term(9), by_term(10)])]
    for scorer in (BM25(), TFIDF(True)):
        for k in (1, 7, 4096):
            run_and_check(L, seg, filters, scorer, k, 4096, 2, sr=sr)
    sr.close()


# ------------------------------------------ the reference's own score-order vectors --

def reference_order_corpus():
    """tests/resources/simple_sequential_order.json of the reference (committed as
    tests/golden/): 8 docs, field "field" = a list of one-character terms.  Returns
    (posting lists per term '0'..'9' as (docs, freqs) with doc id = seq + 1, field lengths)."""
    import json
    rows = json.loads((GOLDEN / "simple_sequential_order.json").read_text())
    assert [r["seq"] for r in rows] == list(range(8))
    lists = []
    for t in "0123456789":
        d = [(r["seq"] + 1, r["field"].count(t)) for r in rows if t in r["field"]]
        lists.append((np.array([x for x, _ in d], np.uint32), np.array([f for _, f in d], np.uint32)))
    lengths = np.array([len(r["field"]) for r in rows], np.uint8)
    assert int(lengths.sum()) == 52          # "TotalFreq = 52", bm25_test.cpp:67-71
    return lists, lengths


# (scorer, norms?, filter terms) -> doc "seq" values in descending score order, exactly as
# the reference's tests assert them (equal scores keep doc order: std::multimap insertion)
REFERENCE_ORDERS = [
    # tests/search/bm25_test.cpp — BM25(k=1.2, b=0.75)
    ("bm25", False, "7", [0, 1, 5, 7]),                 # :557-570 by_term
    ("bm25", False, "8", [3, 7]),                       # :973-992 by_range [8, 9)
    ("bm25", False, "78", [7, 3, 0, 1, 5]),             # :1027-1044 by_range (6, 8]
    ("bm25", False, "678", [7, 0, 5, 3, 2, 1]),         # :1078-1095 by_range [6, 8]
    ("bm25", True, "78", [7, 3, 0, 1, 5]),              # :127-144 with Norm2
    ("bm25", True, "678", [0, 7, 5, 3, 2, 1]),          # :179-197 with Norm2
    # tests/search/tfidf_test.cpp — TFIDF
    ("tfidf", False, "7", [0, 1, 5, 7]),                # :565-568
    ("tfidf", False, "8", [3, 7]),                      # :983-992
    ("tfidf", False, "78", [7, 0, 1, 3, 5]),            # :1035-1043
    ("tfidf", False, "678", [0, 7, 5, 1, 3, 2]),        # :1086-1094
    ("tfidf", True, "78", [7, 0, 3, 1, 5]),             # :133-142 with norms
    ("tfidf", True, "678", [0, 7, 5, 2, 3, 1]),         # :187-196 with norms
]
# two segments (docs 0,2,4,6 | 1,3,5,7), statistics over both: bm25_test.cpp:653-662, 755-775;
# tfidf_test.cpp:655-661, 759-773
REFERENCE_ORDERS_2SEG = [("6", [0, 2, 5]), ("68", [3, 7, 0, 2, 5])]


def _order_scorer(name, with_norms):
    return BM25() if name == "bm25" else TFIDF(with_norms)


def case_reference_score_orders(L):
    """The only score-related vectors the reference's tests hold: the ORDER in which
    bm25_test.cpp / tfidf_test.cpp expect docs of simple_sequential_order.json to be ranked
    (by_range over single-character terms == Or of by_term: every scored term brings its own
    statistics, scores are summed — multiterm_query.cpp).  Checked through the C ABI and
    through the oracle's harness."""
    lists, lengths = reference_order_corpus()
    for layout in (synth.LAYOUT_SCALAR, synth.LAYOUT_SIMD4):
        for name, with_norms, terms, want in REFERENCE_ORDERS:
            seg = synth.segment_from_lists(lists, 8, layout, lengths if with_norms else False)
            seg.total_term_freq = 52
            flt = [Or([by_term(int(t)) for t in terms]) if len(terms) > 1 else by_term(int(terms))]
            scorer = _order_scorer(name, with_norms)
            hits, counts, totals = run_and_check(L, seg, flt, scorer, 8)
            got = [int(d) - 1 for d in hits[0, :int(counts[0])]["doc"]]
            assert got == want, (name, with_norms, terms, got, want)
            ohits, _ = parity.oracle_topk([seg], flt, scorer, 8)[0]
            # the harness heap sorts unstably: compare score classes, then the doc order inside
            key = sorted(zip(-ohits["score"].astype(np.float64), ohits["doc"]))
            assert [int(d) - 1 for _, d in key] == want, ("oracle", name, with_norms, terms)
        # two segments, global statistics, one ranking
        even = [(np.array([(x + 1) // 2 for x in d if x % 2 == 1], np.uint32),
                 np.array([f for x, f in zip(d, fr) if x % 2 == 1], np.uint32)) for d, fr in lists]
        odd = [(np.array([x // 2 for x in d if x % 2 == 0], np.uint32),
                np.array([f for x, f in zip(d, fr) if x % 2 == 0], np.uint32)) for d, fr in lists]
        segs = [synth.segment_from_lists(even, 4, layout, False),
                synth.segment_from_lists(odd, 4, layout, False)]
        segs[0].total_term_freq = int(lengths[0::2].sum())
        segs[1].total_term_freq = int(lengths[1::2].sum())
        for name in ("bm25", "tfidf"):
            scorer = _order_scorer(name, False)
            for terms, want in REFERENCE_ORDERS_2SEG:
                flt = [Or([by_term(int(t)) for t in terms]) if len(terms) > 1 else by_term(int(terms))]
                prep = search.prepare(flt, scorer, [parity.segment_stats(s) for s in segs])
                per_seg = []
                for s in segs:
                    sr = search.SegmentReader.from_synth(s, L=L)
                    b = sr.batch(prep, 8)
                    h, c, _ = b.run().results()
                    per_seg.append((h, c))
                    b.close()
                    sr.close()
                merged = search.merge_topk_host(per_seg, 8)[0]
                got = [2 * (d - 1) + s for _, s, d in merged]      # (segment, local doc) -> seq
                assert got == want, (name, terms, got, want)


def case_boolean_reference_vectors(L, max_doc=2_000_000):
    """tests/golden/boolean_golden.json: the literal posting lists and expected doc ids of
    the reference's disjunction / conjunction / min-match iterator tests
    (tests/search/boolean_filter_tests.cpp, `next` tests) through the C ABI and the oracle."""
    import json
    vectors = json.loads((GOLDEN / "boolean_golden.json").read_text())["vectors"]
    assert len(vectors) >= 30
    ran = 0
    for v in vectors:
        top = max(max(l) for l in v["lists"])
        if top > max_doc:
            continue
        lists = [(np.array(l, np.uint32), np.ones(len(l), np.uint32)) for l in v["lists"]]
        seg = synth.segment_from_lists(lists, top + 1, synth.LAYOUT_SIMD4, norms=False)
        subs = [by_term(i) for i in range(len(lists))]
        flt = {"or": Or(subs), "and": And(subs),
               "minmatch": Or(subs, min_match=v["min_match"])}[v["op"]]
        if v["op"] == "minmatch" and v["min_match"] == 1:
            flt = Or(subs)
        hits, counts, totals = run_and_check(L, seg, [flt], BM25(), 64)
        got = sorted(int(d) for d in hits[0, :counts[0]]["doc"])
        assert got == v["expected"], (v["test"], v["op"], v["min_match"], got, v["expected"])
        assert int(totals[0]) == len(v["expected"])
        ran += 1
    assert ran >= 25


def case_pilot_misled(L, k=600, joined=False):
    """The pilot only sees every 16th doc tile.  Here every high-scoring doc sits in exactly
    the tiles query 0 samples, so the estimated threshold (which extrapolates the sample)
    is far too high; k_select must notice (fewer than k candidates although more docs
    matched) and the batch must re-run with the sound threshold.  Both organisations of the
    doc tiles: work items (4096-doc tiles here) and joined posting streams (12288)."""
    tile, stride, n_tiles = (12288 if joined else 4096), 16, 64
    path = _lib.PATH_JOINED if joined else _lib.PATH_ITEMS
    n_docs = tile * n_tiles
    rng = np.random.default_rng(21)
    sampled = [t for t in range(n_tiles) if t % stride == 0]      # phase of query 0 is 0
    hot = np.concatenate([1 + t * tile + np.sort(rng.choice(tile, 500, replace=False))
                          for t in sampled]).astype(np.uint32)
    # spread over many score bins (a joined stream holds frequencies below 64)
    hot_f = rng.integers(1, 64 if joined else 200, hot.size).astype(np.uint32)
    cold = np.sort(rng.choice(n_docs, 3000, replace=False)).astype(np.uint32) + 1
    lists = [(hot, hot_f), (cold, np.ones(cold.size, np.uint32))]
    seg, sr = open_lists(L, lists, n_docs, synth.LAYOUT_SIMD4, norms=False)
    filters = [by_term(0), Or([by_term(0), by_term(1)]), by_term(1)]
    for scorer in (TFIDF(False), BM25(1.2, 0.0)):
        run_and_check(L, seg, filters, scorer, k, 0 if joined else tile, stride, sr=sr, path=path)
    # the fallback really is what produced those results
    prep = search.prepare(filters, TFIDF(False), [parity.segment_stats(seg)])
    b = sr.batch(prep, k).configure(0 if joined else tile, stride, 0).set_path(path)
    assert b.reruns() == 0
    b.run().results()
    assert b.reruns() == 1
    assert b.path() == path
    b.run().results()            # the batch stays in sound mode: no further re-run
    assert b.reruns() == 1
    b.close()
    sr.close()


def case_paired_tiles(L, num_docs=61_000, max_rank=256, layout=synth.LAYOUT_SIMD4):
    """Joined plain disjunctions on paired doc tiles (16-bit halves pick the docs, k_join_rescore
    forms their exact sums) against the same batch on 32-bit tiles: docs, scores, order, counts
    and totals bit for bit, each also against the oracle.  61 000 docs = 5 tiles: the last pair
    has one tile only.  k = 3 puts the threshold high (few docs pass), k = 1000 low; a batch with
    (set_paired_tiles(2): whatever the segment's size — by default segments below ~2.4 M docs stay on
    32-bit tiles, where the look-ups cost more than the visits save.)"""
    if os.environ.get("IRS_HIP_JOIN_HALF") == "0":
        pytest.skip("IRS_HIP_JOIN_HALF=0: a run of the suite on 32-bit tiles only")
    seg = synth.build_segment(num_docs, max_rank, layout=layout)
    sr = search.SegmentReader.from_synth(seg, L=L)
    ranks = synth.make_queries(10, 8, 2, max_rank, synth.SEED + 15)
    pure = [Or([by_term(int(r) - 1) for r in row]) for row in ranks]
    pure += [by_term(3), by_term(max_rank - 1), Or([by_term(1), by_term(2, 3.0)]),
             Or([by_term(0), by_term(10 * max_rank)]), Or([by_term(0), by_term(1), by_term(2)])]
    for scorer in (BM25(), BM25(1.2, 0.0), TFIDF(True)):
        for k in (3, 100, 1000):
            got = {}
            for paired in (True, False):
                prep = search.prepare(pure, scorer, [parity.segment_stats(seg)])
                b = sr.batch(prep, k).set_path(_lib.PATH_JOINED).set_paired_tiles(2 if paired else 0)
                h, c, t = b.run().results()
                assert b.path() == _lib.PATH_JOINED
                assert b.paired_tiles() == paired, (scorer, k, paired)
                parity.check_single_segment(seg, pure, scorer, k, h, c, t)
                got[paired] = (h.copy(), c.copy(), t.copy())
                # a second run of the same batch (replayed): the same again
                h2, c2, t2 = b.run().results()
                assert np.array_equal(h, h2) and np.array_equal(c, c2) and np.array_equal(t, t2)
                b.close()
            for x, y in zip(got[True], got[False]):
                assert np.array_equal(x, y), (scorer, k)
    # a mixed batch: the plain disjunctions pair, the counting units keep their kernel
    mixed = pure + standard_filters(max_rank, n_or8=1)
    prep = search.prepare(mixed, BM25(), [parity.segment_stats(seg)])
    b = sr.batch(prep, 40).set_path(_lib.PATH_JOINED).set_paired_tiles(2)
    h, c, t = b.run().results()
    assert b.paired_tiles()
    parity.check_single_segment(seg, mixed, BM25(), 40, h, c, t)
    b.close()
    sr.close()


def case_paths_agree(L, num_docs=70_000, max_rank=256, layout=synth.LAYOUT_SIMD4):
    """The two organisations of a disjunction batch — every query decoding its own blocks (work
    items) and the batch decoding every distinct term once (joined posting streams) — return
    the same docs, scores and hit counts, bit for bit (a posting's fixed-point contribution is
    computed by the same arithmetic wherever it is decoded).  Each one is also checked against
    the oracle.  Mixed batches: the And /
    min-match / kMax queries stay on their own kernels while the plain disjunctions join."""
    seg = synth.build_segment(num_docs, max_rank, layout=layout)
    sr = search.SegmentReader.from_synth(seg, L=L)
    ranks = synth.make_queries(6, 8, 2, max_rank, synth.SEED + 5)
    pure = [Or([by_term(int(r) - 1) for r in row]) for row in ranks]
    pure += [by_term(3), by_term(max_rank - 1), Or([by_term(1), by_term(2, 3.0)]),
             Or([by_term(0), by_term(10 * max_rank)])]
    mixed = pure + standard_filters(max_rank, n_or8=1)
    for scorer in (BM25(), BM25(1.2, 0.0), TFIDF(True), TFIDF(False)):
        for filters, k in ((pure, 1000), (mixed, 40)):
            got = {}
            for path in (_lib.PATH_ITEMS, _lib.PATH_JOINED, _lib.PATH_AUTO):
                prep = search.prepare(filters, scorer, [parity.segment_stats(seg)])
                b = sr.batch(prep, k).set_path(path)
                h, c, t = b.run().results()
                # (PATH_AUTO: whichever the cost rule takes for this batch — join_or_pays)
                if path != _lib.PATH_AUTO:
                    assert b.path() == (_lib.PATH_ITEMS if path == _lib.PATH_ITEMS else _lib.PATH_JOINED)
                parity.check_single_segment(seg, filters, scorer, k, h, c, t)
                got[path] = (h.copy(), c.copy(), t.copy())
                b.close()
            hi, ci, ti = got[_lib.PATH_ITEMS]
            hj, cj, tj = got[_lib.PATH_JOINED]
            assert np.array_equal(ci, cj) and np.array_equal(ti, tj)
            # bit for bit: a posting contributes the same fixed-point value on either path —
            # except where the joined path keeps match counts in its accumulators' low bits
            # (And / min-match: contributions rounded to 16 fixed-point units; which of them join
            # without being forced is a cost decision; every run was checked against the oracle)
            for qi, f in enumerate(filters):
                counting = isinstance(f, And) or (isinstance(f, Or) and getattr(f, "min_match", 1) > 1)
                if not counting:
                    assert np.array_equal(hi[qi], hj[qi]), qi
                    assert np.array_equal(got[_lib.PATH_AUTO][0][qi], hj[qi]), qi
    sr.close()


def case_join_counts(L, num_docs=70_000, max_rank=256):
    """Conjunctions and min-match disjunctions as joined posting streams: the number of terms
    holding a doc rides in the low 4 bits of its fixed-point accumulator (join.h COUNT), a doc
    exists when that count reaches the required matches.  Against the oracle, and against the
    block-driven / work-item kernels (same docs wherever the scores are not within the
    tolerance of each other, same hit counts)."""
    seg = synth.build_segment(num_docs, max_rank)
    sr = search.SegmentReader.from_synth(seg, L=L)
    t = by_term
    rare, mid, hot = max_rank - 1, max_rank // 8, 12
    filters = [And([t(hot), t(14)]), And([t(hot), t(mid), t(13)]), And([t(rare), t(hot)]),
               And([t(12), t(13), t(14), t(15)]), And([t(mid), t(mid + 1)]),
               And([t(rare), t(rare - 1), t(rare - 2)]),          # (probably no doc at all)
               Or([t(12), t(13), t(14), t(15)], min_match=2),
               Or([t(12), t(13), t(14), t(15)], min_match=3),
               Or([t(hot), t(mid), t(rare), t(17), t(19)], min_match=4),
               Or([t(12 + i) for i in range(15)], min_match=9),   # 15 terms: the counter's limit
               Or([t(hot), t(mid)], min_match=2),                 # == a conjunction
               Or([t(20), t(21, 2.0), t(22, 0.5)], min_match=2)]
    # ... and what must NOT take that road, in the same batch: an absent term, a single term, 16
    # terms, terms in nearly every doc (idf ~ 0: the rounding would show), a plain disjunction
    others = [And([t(hot), t(10 * max_rank)]), And([t(15)]),
              Or([t(12 + i) for i in range(16)], min_match=9),
              And([t(0), t(1), t(2), t(3)]), Or([t(0), t(1), t(2), t(3)], min_match=3),
              Or([t(hot), t(mid)])]
    for scorer in (BM25(), BM25(1.2, 0.0), TFIDF(False)):
        for fl, k in ((filters, 10), (filters, 1000), (filters + others, 100)):
            got = {}
            for path in (_lib.PATH_ITEMS, _lib.PATH_JOINED):
                prep = search.prepare(fl, scorer, [parity.segment_stats(seg)])
                b = sr.batch(prep, k).set_path(path)
                h, c, tot = b.run().results()
                if fl is filters and not isinstance(scorer, TFIDF):
                    assert b.path() == path   # (only counting queries: they did join)
                parity.check_single_segment(seg, fl, scorer, k, h, c, tot)
                got[path] = (h.copy(), c.copy(), tot.copy())
                b.close()
            assert np.array_equal(got[_lib.PATH_ITEMS][1], got[_lib.PATH_JOINED][1])
            assert np.array_equal(got[_lib.PATH_ITEMS][2], got[_lib.PATH_JOINED][2])
    sr.close()


def case_join_counts_boundary(L, num_docs=70_000, max_rank=256):
    """Counting accumulators AT their eligibility boundary: a joined min-match unit rounds every
    posting to 16 fixed-point units, allowed while `upper / (mean of the need smallest per-term
    minimum scores) <= 125` (count_precise, irs_hip.hip).  The worst case for that rule — every
    doc as long as a 1-byte norm can say (255), two weakly boosted terms that make a doc exist,
    13 strongly boosted ones that set the score's range — with the strong boost bisected to the
    LARGEST value that still joins; just beyond it the unit must fall back to the work items.
    Both against the oracle (1e-5)."""
    seg = synth.build_segment(num_docs, max_rank)
    seg.norms = np.full_like(seg.norms, 255)
    sr = search.SegmentReader.from_synth(seg, L=L)
    t = by_term
    weak = [max_rank // 2, max_rank // 2 + 3]
    strong = [12 + i for i in range(13)]

    def batch(boost):
        flt = [Or([t(w, 1.0) for w in weak] + [t(x, boost) for x in strong], min_match=2)]
        prep = search.prepare(flt, BM25(), [parity.segment_stats(seg)])
        # (joined wherever the unit is ELIGIBLE: the cost rule is not what is probed here)
        return flt, sr.batch(prep, 100).set_path(_lib.PATH_JOINED)

    def joins(boost):
        _, b = batch(boost)
        b.run()
        j = b.path() == _lib.PATH_JOINED
        b.close()
        return j

    lo, hi = 1.0, 1000.0   # (equal boosts join; the strong terms a thousand times the weak ones do not)
    assert joins(lo) and not joins(hi)
    for _ in range(24):
        mid = (lo * hi) ** 0.5
        if joins(mid):
            lo = mid
        else:
            hi = mid
    for boost, want in ((lo, True), (hi, False)):
        flt, b = batch(boost)
        h, c, tot = b.run().results()
        assert (b.path() == _lib.PATH_JOINED) == want, (boost, b.path())
        parity.check_single_segment(seg, flt, BM25(), 100, h, c, tot)
        b.close()
    assert hi / lo < 1.001
    sr.close()


def case_no_norms(L):
    seg = synth.build_segment(20_000, 128)
    seg.norms = None
    sr = search.SegmentReader(seg.doc_file, seg.metas, seg.num_docs, seg.layout, None, 1,
                              seg.docs_with_field, seg.total_term_freq, L=L)
    filters = standard_filters(128, n_or8=2)
    for scorer in (BM25(), TFIDF(True)):
        run_and_check(L, seg, filters, scorer, 50, sr=sr)
    sr.close()


def case_wide_norms(L, width):
    """Norm2 columns wider than one byte (big-endian values, norm.hpp:170-182) take the
    other BM25 expression c0 - c0*c1/(c1 + tf) (bm25.cpp:355-359) and kRSQRT.get<true>."""
    seg = synth.build_segment(20_000, 128)
    rng = np.random.default_rng(width)
    lengths = rng.integers(1, 60_000 if width == 2 else 3_000_000, seg.num_docs)
    be = lengths.astype(">u2" if width == 2 else ">u4")
    seg.norms = np.frombuffer(be.tobytes(), np.uint8).copy()
    seg.norm_width = width
    seg.total_term_freq = int(lengths.sum())
    sr = search.SegmentReader(seg.doc_file, seg.metas, seg.num_docs, seg.layout, seg.norms, width,
                              seg.docs_with_field, seg.total_term_freq, L=L)
    filters = standard_filters(128, n_or8=2)
    for scorer in (BM25(), TFIDF(True)):
        run_and_check(L, seg, filters, scorer, 50, sr=sr)
    sr.close()


def case_full_vocabulary(L, num_docs=30_000, max_rank=1 << 16, n_queries=24, report=None):
    """The shape a real term dictionary has: the whole vocabulary indexed, so that most terms
    are short lists — a vint tail only (df < 128), a single doc (df == 1, nothing in `.doc`),
    or absent in this segment (df == 0) — next to the few long ones.  Disjunctions and
    conjunctions over such terms against the oracle; decode of a spread of them."""
    import time
    t0 = time.perf_counter()
    seg = synth.build_segment(num_docs, max_rank, keep_postings=False)
    t1 = time.perf_counter()
    sr = search.SegmentReader.from_synth(seg, L=L)
    t_open = time.perf_counter() - t1
    dc = np.asarray(seg.metas["docs_count"])
    short = np.nonzero((dc > 1) & (dc < 128))[0]
    single = np.nonzero(dc == 1)[0]
    absent = np.nonzero(dc == 0)[0]
    assert len(short) > max_rank // 2 and len(single) > 0
    rng = np.random.default_rng(3)
    for t in list(rng.choice(short, 20)) + list(rng.choice(single, 5)):
        d, f = sr.decode_term(int(t))
        od, of = oracle.decode_term(seg.doc_file, seg.metas[int(t)], seg.layout)
        assert np.array_equal(d, od) and np.array_equal(f, of), t
    filters = []
    for q in range(n_queries):
        terms = list(rng.choice(short, 6)) + list(rng.choice(single, 1))
        if len(absent):
            terms.append(rng.choice(absent))
        filters.append(Or([by_term(int(t)) for t in terms]))
    filters += [Or([by_term(int(t)) for t in rng.choice(short, 3)] + [by_term(1)]),
                And([by_term(int(rng.choice(short))), by_term(2)]),
                And([by_term(int(rng.choice(single))), by_term(0)])]
    t2 = time.perf_counter()
    run_and_check(L, seg, filters, BM25(), 50, sr=sr)
    if report is not None:
        report.update(terms=len(dc), short=len(short), single=len(single), absent=len(absent),
                      build_s=t1 - t0, open_s=t_open, query_and_check_s=time.perf_counter() - t2,
                      device_mb=sr.device_bytes() / 1e6)
    sr.close()


def case_legacy_norms(L):
    """The legacy `Norm` feature (norm.hpp:46-70: float 1/sqrt(|doc|) per doc): BM25 takes
    tf = sqrt(freq) and norm = 1/stored through c0 - c0*c1/(c1 + tf) (bm25.cpp:333-359,
    242-249), TF-IDF multiplies by the stored value (tfidf.cpp:214-219) — boolean queries,
    conjunctions (incl. WAND) and phrases."""
    seg = synth.build_segment(20_000, 128, with_positions=True)
    lengths = seg.norms.astype(np.float32)
    seg.norms = np.frombuffer((np.float32(1.0) / np.sqrt(lengths)).astype("<f4").tobytes(),
                              np.uint8).copy()
    seg.norm_width = oracle.NORM_LEGACY_F32   # what parity.oracle_view hands to the oracle
    sr = search.SegmentReader(seg.doc_file, seg.metas, seg.num_docs, seg.layout, seg.norms, 4,
                              seg.docs_with_field, seg.total_term_freq, L=L,
                              pos_file=seg.pos_file, norm_kind=1)
    filters = standard_filters(128, n_or8=2)
    for scorer in (BM25(), TFIDF(True)):
        h0, c0, t0 = run_and_check(L, seg, filters, scorer, 50, sr=sr)
        wb = sr.batch(search.prepare(filters, scorer, [parity.segment_stats(seg)]), 50).set_wand(True)
        h1, c1, t1 = wb.run().results()
        wb.close()
        assert np.array_equal(c0, c1)
        for q in range(len(filters)):
            assert np.array_equal(h0[q, :int(c0[q])], h1[q, :int(c0[q])]), q
        run_phrases(L, seg, [by_phrase([0, 1]), by_phrase([3, 9, 0])], scorer, 20, sr=sr)
    sr.close()


def case_zero_boost(L):
    """A boost of 0 is legal (by_term::boost, filter.hpp): the term still matches, its
    postings score 0 — alone (every score is 0, ranked by doc id) and next to boosted terms."""
    seg = synth.build_segment(20_000, 128)
    sr = search.SegmentReader.from_synth(seg, L=L)
    filters = [by_term(5, boost=0.0),
               Or([by_term(3, boost=0.0), by_term(40), by_term(77, boost=0.0)]),
               And([by_term(2, boost=0.0), by_term(9)]),
               Or([by_term(1), by_term(6, boost=0.0), by_term(30)], min_match=2),
               Or([by_term(8, boost=0.0), by_term(21, boost=0.0)])]
    for scorer in (BM25(), TFIDF(False)):
        hits, counts, totals = run_and_check(L, seg, filters, scorer, 30, sr=sr)
        assert (hits[0, :int(counts[0])]["score"] == 0).all()
        assert (hits[4, :int(counts[4])]["score"] == 0).all()
        d = hits[0, :int(counts[0])]["doc"]
        assert (d[:-1] < d[1:]).all()
    sr.close()


def case_merge_types(L):
    """boolean_filter::merge_type() = kMax / kMin (scorer.hpp:399-423): Or, And and min-match,
    with the reference's own kMin-in-a-disjunction behaviour (min of two where both match and 0
    elsewhere; 0 everywhere for three or more: the zeroed score buffer), every scorer family,
    both accumulator widths, dense and sparse terms (wide doc ranges in the conjunction)."""
    from iresearch_amd.search import MERGE_MAX, MERGE_MIN
    seg = synth.build_segment(60_000, 512)
    sr = search.SegmentReader.from_synth(seg, L=L)
    t = by_term
    filters = []
    for mg in (MERGE_MAX, MERGE_MIN):
        filters += [Or([t(3), t(40), t(77), t(200)], merge=mg),
                    Or([t(5), t(9)], merge=mg),
                    Or([t(2, 3.0), t(30, 0.5)], merge=mg),
                    And([t(1), t(6), t(30)], merge=mg),
                    And([t(4), t(300)], merge=mg),
                    And([t(0), t(2), t(3), t(5), t(7), t(11)], merge=mg),
                    Or([t(1), t(6), t(30), t(12)], min_match=2, merge=mg),
                    Or([t(1), t(6), t(30)], min_match=3, merge=mg),
                    Or([t(7), t(10 ** 6)], merge=mg),        # one side absent: a single iterator
                    Or([t(8)], merge=mg)]
    for scorer in (BM25(), TFIDF(True), BM25(1.2, 0.0)):
        hits, counts, totals = run_and_check(L, seg, filters, scorer, 50, sr=sr)
        n = len(filters) // 2
        # kMin, Or of four: every score is 0
        assert (hits[n, :int(counts[n])]["score"] == 0).all() and counts[n] == 50
        # kMin, Or of two: some docs hold both terms (score > 0), the others score 0
        s = hits[n + 1, :int(counts[n + 1])]["score"]
        assert s[0] > 0
    import os
    os.environ["IRS_HIP_ACC"] = "64"
    try:
        run_and_check(L, seg, filters, BM25(), 50, sr=sr)
    finally:
        del os.environ["IRS_HIP_ACC"]
    sr.close()


def case_accumulator_switch(L):
    """32-bit fixed-point accumulators are used while U / (smallest possible posting score)
    <= 1000 for every query of the batch, 64-bit ones beyond (irs_hip.hip batch_create).  Batches
    sitting just below and just above that switch, on the docs that make the bound tight — the
    longest docs (norm 255), tf = 1, matched by the LOW-boost term only — must both stay within
    1e-5 of the oracle for every returned doc, the smallest scores included (k takes every
    match)."""
    n_docs = 6000
    rng = np.random.default_rng(31)
    norms = np.full(n_docs, 255, np.uint8)
    norms[::7] = 40
    da = np.sort(rng.choice(np.arange(1, n_docs + 1), 1500, replace=False)).astype(np.uint32)
    db = np.sort(rng.choice(np.arange(1, n_docs + 1), 700, replace=False)).astype(np.uint32)
    dc = np.sort(rng.choice(np.arange(1, n_docs + 1), 300, replace=False)).astype(np.uint32)
    lists = [(da, np.ones(da.size, np.uint32)),
             (db, rng.integers(1, 6, db.size).astype(np.uint32)),
             (dc, rng.integers(1, 30, dc.size).astype(np.uint32))]
    seg, sr = open_lists(L, lists, n_docs, synth.LAYOUT_SIMD4, norms)
    stats = [parity.segment_stats(seg)]
    scorer = BM25()
    # U / min-score as batch_create computes it, for an Or of (term 0, boost 1) and (term 1, boost B)
    c0 = [p.scorers[0][1] for p in search.prepare([by_term(0), by_term(1), by_term(2)], scorer, stats)]
    nc, nl = search.prepare([by_term(0)], scorer, stats)[0].scorers[0][2:4]
    smin = c0[0] - c0[0] / (1.0 + 1.0 / (nc + nl * 255.0))

    def boost_for(ratio):   # (the batch's widest query: terms 0, 1 at boost B, 2 at B / 2)
        return (ratio * smin - c0[0]) / (c0[1] + c0[2] / 2.0)
    for ratio, want_joined in ((900.0, True), (990.0, True), (1010.0, False), (1500.0, False)):
        B = boost_for(ratio)
        filters = [Or([by_term(0), by_term(1, B)]), Or([by_term(0), by_term(1, B), by_term(2, B / 2)]),
                   Or([by_term(1, B), by_term(0)])]
        prep = search.prepare(filters, scorer, stats)
        b = sr.batch(prep, 4096)
        hits, counts, totals = b.run().results()
        # (which side of the switch the batch fell on shows in the path it could take)
        assert (b.path() == _lib.PATH_JOINED) == want_joined, (ratio, b.path())
        parity.check_single_segment(seg, filters, scorer, 4096, hits, counts, totals)
        assert counts[0] == totals[0] == np.union1d(da, db).size     # every match returned
        lo = hits[0, int(counts[0]) - 1]
        assert lo["score"] > 0 and lo["score"] <= smin * (1 + 1e-4)  # ... down to the smallest one
        b.close()
    sr.close()


def case_plan_ahead(L):
    """irs_hip_batch_plan: a run whose planning stage was queued ahead returns exactly what a
    plain run returns — tile batches, conjunctions, phrases, repeated and interleaved with
    plain runs."""
    seg = synth.build_segment(40_000, 128, with_positions=True)
    sr = search.SegmentReader.from_synth(seg, L=L)
    st = [parity.segment_stats(seg)]
    for filters in ([Or([by_term(t) for t in (3, 17, 40, 99)]), by_term(5),
                     And([by_term(2), by_term(9)]), Or([by_term(1), by_term(6), by_term(30)], min_match=2)],
                    [by_phrase([1, 2]), by_phrase([0, 3, 5])]):
        prep = search.prepare(filters, BM25(), st)
        b = sr.batch(prep, 25)
        ref = [x.copy() for x in b.run().results()]
        for _ in range(2):
            got = b.plan().run().results()
            assert all(np.array_equal(a, g) for a, g in zip(ref, got))
        got = b.run().results()
        assert all(np.array_equal(a, g) for a, g in zip(ref, got))
        got = b.plan().plan().run().results()     # planning twice is harmless
        assert all(np.array_equal(a, g) for a, g in zip(ref, got))
        # a plan queued ahead, then the batch is re-dealt (the setters drop the plan's tables and
        # reallocate: they wait for the queued plan first — plan_pending), then run
        for path in (_lib.PATH_ITEMS, _lib.PATH_AUTO):
            got = b.plan().set_path(path).run().results()
            assert all(np.array_equal(a, g) for a, g in zip(ref[1:], got[1:]))   # (counts, totals)
        got = b.plan().configure(0, 8, 0).plan().run().results()
        assert all(np.array_equal(a, g) for a, g in zip(ref[1:], got[1:]))
        b.close()
        # ... and a plan queued, the batch reconfigured and destroyed without ever running it
        b = sr.batch(prep, 25).plan()
        b.configure(0, 4, 0)
        b.close()
        b = sr.batch(prep, 25).plan().plan()
        b.close()
    sr.close()


def case_fresh_batches_and_trim(L):
    """The life cycle bench.py times: a NEW batch per step (created, run, read, destroyed), its
    buffers coming from and going back to the library's pool — results never depend on what an
    earlier batch left in a recycled buffer (the emulator poisons freed blocks) — with
    irs_hip_device_trim handing the pool back in between, and a batch destroyed without ever
    having run."""
    seg = synth.build_segment(60_000, 256)
    sr = search.SegmentReader.from_synth(seg, L=L)
    st = [parity.segment_stats(seg)]
    sets = []
    for i in range(4):
        ranks = synth.make_queries(6 + 3 * i, 2 + 2 * i, 2, 256, synth.SEED + 40 + i)
        filters = [Or([by_term(int(r) - 1) for r in row]) for row in ranks]
        filters += [And([by_term(3 + i), by_term(40)]), Or([by_term(5), by_term(7 + i), by_term(90)], min_match=2)]
        sets.append((filters, search.prepare(filters, BM25(), st)))
    ref = []
    for filters, prep in sets:
        b = sr.batch(prep, 50)
        ref.append([x.copy() for x in b.run().results()])
        parity.check_single_segment(seg, filters, BM25(), 50, *ref[-1])
        b.close()
    for rnd in range(3):
        pending = None
        for i in (2, 0, 3, 1, 1, 3):
            b = sr.batch(sets[i][1], 50).run()
            if pending is not None:   # (the previous batch is read after the next one was queued)
                j, pb = pending
                got = pb.results()
                assert all(np.array_equal(a, g) for a, g in zip(ref[j], got)), (rnd, j)
                pb.close()
            pending = (i, b)
        j, pb = pending
        assert all(np.array_equal(a, g) for a, g in zip(ref[j], pb.results())), (rnd, j)
        pb.close()
        sr.batch(sets[rnd][1], 50).close()     # created, never run
        _lib.check(L, L.irs_hip_device_trim(0), "irs_hip_device_trim")
    sr.close()


def case_host_results(L):
    """irs_hip_batch_results_to_host / _host_results: the checked results in the batch's own
    page-locked memory, queued behind the batch's run only — the copy of batch i is requested after
    batch i + 1 was handed to the library's worker thread (irs_hip_batch_run returns before the
    run is queued) — equal what irs_hip_batch_results returns, single- and multi-segment, also
    after a re-run (estimated threshold misled: the copy follows the repeated run)."""
    seg = synth.build_segment(60_000, 256)
    seg2 = synth.build_segment(9_000, 256, first_doc=60_000)
    sr, sr2 = (search.SegmentReader.from_synth(x, L=L) for x in (seg, seg2))
    st = [parity.segment_stats(seg), parity.segment_stats(seg2)]
    sets = []
    for i in range(3):
        ranks = synth.make_queries(5 + 2 * i, 3 + 2 * i, 2, 256, synth.SEED + 70 + i)
        filters = [Or([by_term(int(r) - 1) for r in row]) for row in ranks]
        filters += [And([by_term(3 + i), by_term(40)]), by_term(200 + i)]
        sets.append(search.prepare(filters, BM25(), st))
    for readers in (sr, [sr, sr2]):
        ref = []
        for prep in sets:
            b = search.QueryBatch(readers, prep, 40)
            ref.append([x.copy() for x in b.run().results()])
            b.close()
        pending = None
        for i in (0, 2, 1, 1, 0):
            b = search.QueryBatch(readers, sets[i], 40).run()
            if pending is not None:
                j, pb = pending
                got = pb.results_to_host().host_results()
                assert all(np.array_equal(a, g) for a, g in zip(ref[j], got)), j
                # (again: a second copy waits for the first)
                got = pb.results_to_host().results_to_host().host_results()
                assert all(np.array_equal(a, g) for a, g in zip(ref[j], got)), j
                pb.close()
            pending = (i, b)
        j, pb = pending
        got = pb.results_to_host().host_results()
        assert all(np.array_equal(a, g) for a, g in zip(ref[j], got)), j
        pb.close()
    # no copy requested: EINVAL, not stale memory
    b = sr.batch(sets[0], 40).run()
    hp = C.c_void_p()
    assert L.irs_hip_batch_host_results(b.handle, C.byref(hp), None, None, None) == _lib.EINVAL
    b.close()
    sr.close()
    sr2.close()


def case_min_score_pushdown(L):
    """irs::score::Min (score_function.hpp:42-142; the harness pushes its heap's k-th score,
    index-search.cpp:737-777): with the k-th score of a first run as threshold the same top-k
    comes back; with the score of rank 10 at least those 10, all of them a prefix of the first
    list, and never a doc below the threshold; hit counts are unchanged.  Or, And, phrase."""
    seg = synth.build_segment(50_000, 256, with_positions=True)
    sr = search.SegmentReader.from_synth(seg, L=L)
    st = [parity.segment_stats(seg)]
    sets = [[Or([by_term(t) for t in (3, 17, 40, 99)]), Or([by_term(1), by_term(5)]),
             And([by_term(2), by_term(9)]), And([by_term(0), by_term(4), by_term(30)]),
             by_term(12)],
            [by_phrase([1, 2]), by_phrase([0, 3])]]
    k = 40
    for filters in sets:
        prep = search.prepare(filters, BM25(), st)
        b = sr.batch(prep, k)
        h0, c0, t0 = (x.copy() for x in b.run().results())
        assert (c0 >= 12).all()
        kth = np.array([h0[q, c0[q] - 1]["score"] for q in range(len(filters))], np.float32)
        h1, c1, t1 = (x.copy() for x in b.set_min_scores(kth).run().results())
        assert np.array_equal(c0, c1) and np.array_equal(t0, t1) and np.array_equal(h0, h1)
        tenth = np.array([h0[q, 9]["score"] for q in range(len(filters))], np.float32)
        h2, c2, t2 = (x.copy() for x in b.set_min_scores(tenth).run().results())
        assert np.array_equal(t0, t2) and b.reruns() == 0
        for q in range(len(filters)):
            n = int(c2[q])
            # exactly the docs at or above the threshold (ties with the 10th included)
            assert n == int((h0[q, :int(c0[q])]["score"] >= tenth[q]).sum()) >= 10
            assert np.array_equal(h2[q, :n], h0[q, :n])
        # a threshold nothing reaches: no hits returned, the matches still counted
        h3, c3, t3 = b.set_min_scores(np.full(len(filters), 1e30, np.float32)).run().results()
        assert np.array_equal(t0, t3) and (c3 == 0).all()
        h4, c4, t4 = b.set_min_scores(None).run().results()
        assert np.array_equal(h0, h4) and np.array_equal(c0, c4)
        b.close()
    sr.close()


def case_header_chain(L, layout, n_docs=120_000):
    """The directory kernels follow the chain of block headers window by window (chain_orbit,
    kernels.h: pointer doubling over "the next header if one starts here"): lists long enough for
    many LDS windows whose headers take every path — ALL_EQUAL runs with a one-byte value and
    with a two-byte value, packed blocks from 1 to 32 bits, more headers in a window than one
    link list holds, mixed — for `.doc` with and without frequencies and for `.pos`.
    Checked through the decoders and the directory against the lists and the oracle."""
    rng = np.random.default_rng(23)
    N = n_docs

    def some(n):
        return np.sort(rng.choice(np.arange(1, N + 1), n, replace=False)).astype(np.uint32)

    every = np.arange(1, N + 1, dtype=np.uint32)                 # delta 1 throughout
    stride = np.arange(1, N + 1, 300, dtype=np.uint32)[:128 * 3 + 5]   # delta 300: 2-byte value
    lists = [
        (every, np.full(every.size, 3, np.uint32)),               # both parts ALL_EQUAL
        (every, rng.integers(1, 4, every.size).astype(np.uint32)),   # ALL_EQUAL docs, 2-bit freqs
        (every[::2].copy(), np.full(every.size // 2, 200, np.uint32)),  # freqs ALL_EQUAL, 2 bytes
        (stride, np.ones(stride.size, np.uint32)),
        (some(N // 3), rng.integers(1, 60, N // 3).astype(np.uint32)),
        (some(5000), rng.integers(1, 2**20, 5000).astype(np.uint32)),   # wide freq blocks
    ]
    # wide doc deltas: a big segment id space would be needed for 32 bits; 17+ bits here
    wide = np.cumsum(rng.integers(1, 2 * N // 700, 700)).astype(np.uint32)
    lists.append((wide[wide <= N], np.ones(int((wide <= N).sum()), np.uint32)))
    seg, sr = open_lists(L, lists, N, layout)
    for t, (d, f) in enumerate(lists):
        assert_decode(sr, seg, t, d, f)
        last, offs = sr.term_directory(t)
        nb = d.size // 128
        assert np.array_equal(last[:nb], d[127::128][:nb]), ("last docs", t)
        assert (np.diff(offs[:nb].astype(np.int64)) > 0).all()
    sr.close()
    # a field without frequencies: one header per block
    nseg = synth.segment_from_lists([(d, None) for d, _ in lists], N, layout, norms=False)
    nsr = search.SegmentReader.from_synth(nseg, L=L, has_freq=False)
    for t, (d, _) in enumerate(lists):
        got, _ = nsr.decode_term(t, want_freq=False)
        od, _ = oracle.decode_term(nseg.doc_file, nseg.metas[t], layout, want_freq=False,
                                   field_has_freq=False)
        assert np.array_equal(got, d) and np.array_equal(od, d), ("no freq", t)
    nsr.close()
    # positions: constant deltas (ALL_EQUAL pos blocks, one- and two-byte values), 1..7-bit
    # packed blocks over many windows, and wide ones
    n1 = 40_000
    d1 = some(n1)
    f1 = np.full(n1, 4, np.uint32)
    plists = [
        (d1, f1, np.tile(np.array([2, 4, 6, 8], np.uint32), n1)),
        (d1[:3000], f1[:3000], np.tile(np.array([200, 400, 600, 800], np.uint32), 3000)),
    ]
    f2 = rng.integers(1, 9, n1).astype(np.uint32)
    plists.append((d1, f2, np.concatenate([np.cumsum(rng.integers(1, 100, int(k))) for k in f2])
                   .astype(np.uint32)))
    f3 = rng.integers(1, 3, 2000).astype(np.uint32)
    plists.append((d1[:2000], f3, np.concatenate([np.cumsum(rng.integers(1, 2**28, int(k)))
                                                  for k in f3]).astype(np.uint32)))
    pseg = synth.segment_from_lists(plists, N, layout)
    psr = search.SegmentReader.from_synth(pseg, L=L)
    for t, (d, f, p) in enumerate(plists):
        assert_decode(psr, pseg, t, d, f)
        got = psr.decode_positions(t)
        assert np.array_equal(got, p), ("positions", layout, t)
        assert np.array_equal(got, oracle.decode_positions(pseg.doc_file, pseg.pos_file,
                                                           pseg.metas[t], layout)), t
    psr.close()
    # skip lists longer than one window / one entry list of k_wand_skip0 (937 entries, 1 and 3
    # scorers): the pairs taken from the index are the oracle's
    norms = rng.integers(40, 256, N).astype(np.uint8)
    half = some(N // 2)
    wl = [(every, np.minimum(rng.integers(1, 30, N), norms).astype(np.uint32)),
          (half, np.minimum(rng.integers(1, 200, half.size), norms[half - 1]).astype(np.uint32))]
    for kinds in ([synth.WAND_MIN_NORM], [synth.WAND_MAX_FREQ, synth.WAND_DIV_NORM, synth.WAND_MIN_NORM]):
        wseg = synth.segment_from_lists(wl, N, layout, norms, wand_kinds=kinds)
        wsr = search.SegmentReader.from_synth(wseg, L=L)
        for t, (d, f) in enumerate(wl):
            sl, sp, levels, mf, nm = oracle.read_skip0(wseg.doc_file, wseg.metas[t], len(kinds), True)
            assert len(sl) == d.size // 128 - (1 if d.size % 128 == 0 else 0) and levels >= 2
            gmf, gmn = wsr.term_blockmax(t)
            assert np.array_equal(gmf[:len(sl)], mf) and np.array_equal(gmn[:len(sl)], nm), (kinds, t)
            assert np.array_equal(gmf, f[:128 * len(gmf)].reshape(-1, 128).max(axis=1))
        from_index, total = wsr.wand_source()
        assert from_index == sum(d.size // 128 - (1 if d.size % 128 == 0 else 0) for d, _ in wl)
        wsr.close()


def case_decode_without_freq(L, layout):
    """Iterator requested without IndexFeatures::FREQ on a FREQ field: freq blocks are
    skipped (formats_10.cpp:1746-1750) — docs must be identical."""
    seg = synth.build_segment(8_000, 64, layout=layout, keep_postings=True)
    sr = search.SegmentReader.from_synth(seg, L=L)
    for r in (1, 2, 17, 40, 64):
        d, f = sr.decode_term(r - 1, want_freq=False)
        assert f is None and np.array_equal(d, seg.postings[r][0])
        od, _ = oracle.decode_term(seg.doc_file, seg.metas[r - 1], layout, want_freq=False)
        assert np.array_equal(d, od)
    sr.close()


def case_wand_data(L, layout):
    """Fields indexed WITH scorers (formats 1_4/1_5): "wand data" sits in front of short
    lists' tails, in front of the skip levels and in every skip entry (formats_10.cpp:686-688,
    :778, :990-999).  Decode, block directory, bit_union and query results must be what they
    are for the same lists written without scorers."""
    rng = np.random.default_rng(5)
    n_docs = 40_000
    norms = rng.integers(40, 256, n_docs).astype(np.uint8)
    sizes = (1, 2, 90, 127, 128, 129, 255, 256, 1500, 9000)
    lists = []
    for n in sizes:
        d = np.sort(rng.choice(n_docs, n, replace=False)).astype(np.uint32) + 1
        f = np.minimum(rng.integers(1, 30, n), norms[d - 1]).astype(np.uint32)
        lists.append((d, f))
    plain = synth.segment_from_lists(lists, n_docs, layout, norms)
    for kinds in ([synth.WAND_MIN_NORM], [synth.WAND_MAX_FREQ, synth.WAND_DIV_NORM, synth.WAND_MIN_NORM]):
        seg = synth.segment_from_lists(lists, n_docs, layout, norms, wand_kinds=kinds)
        wc = len(kinds)
        assert seg.wand_count == wc and seg.doc_file.size > plain.doc_file.size
        sr = search.SegmentReader.from_synth(seg, L=L)
        for t, (d, f) in enumerate(lists):
            gd, gf = sr.decode_term(t)
            od, of = oracle.decode_term(seg.doc_file, seg.metas[t], layout, wand_count=wc)
            assert np.array_equal(gd, d) and np.array_equal(gf, f), t
            assert np.array_equal(od, d) and np.array_equal(of, f), t
            last, offs = sr.term_directory(t)
            sl, sp, levels, mf, nm = oracle.read_skip0(seg.doc_file, seg.metas[t], wc, True)
            assert np.array_equal(last[:len(sl)], sl)
            assert np.array_equal(offs[1:], sp[:max(len(offs) - 1, 0)])
            # scorer 0's payload of every skip entry: the block's max freq [and min norm]
            for b in range(len(sl)):
                blk = slice(128 * b, 128 * b + 128)
                assert mf[b] == f[blk].max(), (t, b)
                if kinds[0] == synth.WAND_MIN_NORM:
                    assert nm[b] == max(int(norms[d[blk] - 1].min()), int(f[blk].max())), (t, b)
            # the block-max data the GPU prunes with == what the index's own wand data says
            # (FreqNormProducer stores max(min norm, max freq), wand_writer.hpp:198-209); the
            # GPU also has it for a list's last full block, which has no skip entry
            gmf, gmn = sr.term_blockmax(t)
            assert len(gmf) == len(d) // 128
            for b in range(len(gmf)):
                blk = slice(128 * b, 128 * b + 128)
                assert gmf[b] == f[blk].max(), (t, b)
                if b < len(sl):      # read from the index's skip entry (k_wand_skip0); a
                    # frequency-only payload reads as norm == freq (FreqNormSource::Read)
                    assert gmf[b] == mf[b] and gmn[b] == nm[b], (t, b)
                else:                # no skip entry for a list's last block: derived (k_block_max)
                    assert gmn[b] == norms[d[blk] - 1].min(), (t, b)
        # where the pairs came from: one skip entry per full block but the last of every list
        from_index, total = sr.wand_source()
        assert total == sum(len(d) // 128 for d, _ in lists)
        assert from_index == sum(max(len(d) // 128 - (1 if len(d) % 128 == 0 else 0), 0)
                                 for d, _ in lists if len(d) > 128), (from_index, total)
        n_words = (n_docs + 64) // 64
        terms = list(range(len(lists)))
        got, cnt = sr.bit_union(terms, n_words)
        want, ocnt = oracle.bit_union(seg.doc_file, [seg.metas[t] for t in terms], layout, True,
                                      n_words, wand_count=wc)
        assert cnt == ocnt and np.array_equal(got, want)
        # same query results as on the segment written without scorers
        filters = [by_term(2), by_term(0), Or([by_term(t) for t in range(10)]),
                   And([by_term(8), by_term(9)]), Or([by_term(2), by_term(3)], min_match=2)]
        h1, c1, t1 = run_and_check(L, seg, filters, BM25(), 50, sr=sr)
        h0, c0, t0 = run_and_check(L, plain, filters, BM25(), 50)
        assert np.array_equal(h0, h1) and np.array_equal(c0, c1) and np.array_equal(t0, t1)
        sr.close()
    # an index written without scorers has no pairs of its own: all derived
    sr = search.SegmentReader.from_synth(plain, L=L)
    assert sr.wand_source()[0] == 0
    sr.close()
    # scorer 0 was a DivNorm producer (TFIDF with norms, BM11): its pair is the (freq, norm) of
    # the doc with the largest freq / norm — no bound for BM25 or a MaxFreq scorer (the
    # reference refuses the combination, Scorer::compatible scorer.cpp:46-49).  The index's
    # pairs must NOT be taken; pruning with the derived ones returns the exhaustive top k.
    seg = synth.segment_from_lists(lists, n_docs, layout, norms,
                                   wand_kinds=[synth.WAND_DIV_NORM, synth.WAND_MIN_NORM])
    assert seg.wand_type == _lib.WAND_DIV_NORM
    sr = search.SegmentReader.from_synth(seg, L=L)
    assert sr.wand_source()[0] == 0
    for t, (d, f) in enumerate(lists):
        gmf, gmn = sr.term_blockmax(t)
        for b in range(len(gmf)):
            blk = slice(128 * b, 128 * b + 128)
            assert gmf[b] == f[blk].max() and gmn[b] == norms[d[blk] - 1].min(), (t, b)
    filters = [by_term(9), Or([by_term(8), by_term(9)]), And([by_term(8), by_term(9)]),
               And([by_term(7), by_term(9), by_term(8)])]
    for scorer in (BM25(), TFIDF(False), BM25(1.2, 0.0)):
        prep = search.prepare(filters, scorer, [parity.segment_stats(seg)])
        ex = sr.batch(prep, 10).set_path(_lib.PATH_ITEMS)
        h0, c0, t0 = ex.run().results()
        ex.close()
        wb = sr.batch(prep, 10).set_wand(True)
        h1, c1, t1 = wb.run().results()
        wb.close()
        assert np.array_equal(c0, c1) and np.array_equal(h0, h1)
        parity.check_single_segment(seg, filters, scorer, 10, h0, c0, t0)
    sr.close()
    # the whole-corpus builder writes the same framing
    a = synth.build_segment(6_000, 64, layout=layout, keep_postings=True, wand_count=2)
    sr = search.SegmentReader.from_synth(a, L=L)
    for r in (1, 7, 30, 64):
        gd, gf = sr.decode_term(r - 1)
        assert np.array_equal(gd, a.postings[r][0]) and np.array_equal(gf, a.postings[r][1])
    assert sr.wand_source()[0] > 0
    sr.close()
    # ... and with positions the skip entries also carry the `.pos` fields (ReadState :1063-1080):
    # the pairs read from them are the oracle's
    p = synth.build_segment(20_000, 32, layout=layout, wand_count=1, with_positions=True)
    sr = search.SegmentReader.from_synth(p, L=L)
    from_index, total = sr.wand_source()
    assert 0 < from_index <= total
    for t in (0, 3, 17):
        sl, sp, levels, mf, nm = oracle.read_skip0(p.doc_file, p.metas[t], 1, True, has_pos=True)
        gmf, gmn = sr.term_blockmax(t)
        assert len(sl) > 0
        assert np.array_equal(gmf[:len(sl)], mf) and np.array_equal(gmn[:len(sl)], nm), t
    sr.close()


def case_bit_union(L, layout, has_freq=True):
    """postings_reader::bit_union (formats_10.cpp:3716-3806): bit-exact bitset and the
    reference's return value (sum of docs_count), over single-doc terms, tail-only terms,
    exact multiples of 128, all-equal blocks, duplicates and pre-set bits."""
    rng = np.random.default_rng(99)
    n_docs = 70_000
    lists = [
        np.array([7], np.uint32),                                        # single doc
        np.sort(rng.choice(n_docs, 100, replace=False)).astype(np.uint32) + 1,   # tail only
        np.arange(1, 129, dtype=np.uint32),                              # one ALL_EQUAL block
        np.arange(5, 5 + 3 * 384, 3, dtype=np.uint32),                   # 3 ALL_EQUAL blocks
        np.sort(rng.choice(n_docs, 128 * 40 + 77, replace=False)).astype(np.uint32) + 1,
        np.sort(rng.choice(n_docs, 128 * 9, replace=False)).astype(np.uint32) + 1,
        np.array([n_docs], np.uint32),                                   # last representable doc
    ]
    freqs = [rng.integers(1, 9, l.size).astype(np.uint32) if has_freq else None for l in lists]
    seg = synth.segment_from_lists(list(zip(lists, freqs)), n_docs, layout, norms=False)
    sr = search.SegmentReader.from_synth(seg, L=L, has_freq=has_freq)
    if not has_freq:  # the no-FREQ framing also goes through the plain decoder
        for t, l in enumerate(lists):
            d, _ = sr.decode_term(t, want_freq=False)
            od, _ = oracle.decode_term(seg.doc_file, seg.metas[t], layout, want_freq=False,
                                       field_has_freq=False)
            assert np.array_equal(d, l) and np.array_equal(od, l), t
    n_words = (n_docs + 1 + 63) // 64
    for terms in ([0], [1, 2], [4], list(range(len(lists))), [4, 4, 0], []):
        init = np.zeros(n_words, np.uint64)
        init[3] = np.uint64(0x8000000000000001)   # bits already set by the caller survive
        got, cnt = sr.bit_union(terms, n_words, init)
        want, ocnt = oracle.bit_union(seg.doc_file, [seg.metas[t] for t in terms], layout,
                                      has_freq, n_words, init)
        assert cnt == ocnt == sum(lists[t].size for t in terms)
        assert np.array_equal(got, want), terms
        expect = init.copy()
        for t in terms:
            np.bitwise_or.at(expect, lists[t] // 64, np.uint64(1) << (lists[t] % 64).astype(np.uint64))
        assert np.array_equal(got, expect), terms
    # several unions in one call, only their populations coming back (irs_hip_bit_union_counts)
    sets = [[0], [1, 2], [4], list(range(len(lists))), [4, 4, 0], [], [0xFFFFFFFF, 3]]
    pops = sr.bit_union_counts(sets)
    for terms, pop in zip(sets, pops):
        want, _ = oracle.bit_union(seg.doc_file, [seg.metas[t] for t in terms if t != 0xFFFFFFFF], layout,
                                   has_freq, n_words)
        assert int(pop) == int(np.unpackbits(want.view(np.uint8)).sum()), terms
    # a bitset too short for the segment: docs beyond it are dropped, nothing is written past it
    got, cnt = sr.bit_union([4], 100)
    want, _ = oracle.bit_union(seg.doc_file, [seg.metas[4]], layout, has_freq, 100)
    assert np.array_equal(got, want) and cnt == lists[4].size
    sr.close()


def case_multi_segment(L, num_docs, max_rank, n_segs=3, k=100, device_merge=True):
    """Segments are independent units with private doc ids; statistics are global
    (term_filter.cpp:102-125); one heap over all segments (index-search.cpp:719-779)."""
    per = num_docs // n_segs
    segs = [synth.build_segment(per, max_rank, first_doc=i * per) for i in range(n_segs)]
    # a term missing from one segment
    segs[1].metas[max_rank - 1]["docs_count"] = 0
    readers = [search.SegmentReader.from_synth(s, L=L) for s in segs]
    filters = standard_filters(max_rank, n_or8=3)
    scorer = BM25()
    prep = search.prepare(filters, scorer, [parity.segment_stats(s) for s in segs])
    per_seg = []
    batches = []
    for s, r in zip(segs, readers):
        b = r.batch(prep, k)
        hits, counts, totals = b.run().results()
        parity.check_single_segment(s, filters, scorer, k, hits, counts, totals, segs)
        per_seg.append((hits, counts))
        batches.append(b)
    merged = search.merge_topk_host(per_seg, k)
    ref = parity.oracle_topk(segs, filters, scorer, k)
    for q, (rows, (ohits, total)) in enumerate(zip(merged, ref)):
        assert len(rows) == len(ohits), q
        if not rows:
            continue
        # same score multiset as the reference harness (ties may pick other docs)
        a = np.array([r[0] for r in rows], np.float32)
        o = np.sort(ohits["score"])[::-1]
        assert np.allclose(a, o, rtol=parity.REL_TOL, atol=0), q
    if device_merge:
        import torch
        from iresearch_amd import distributed
        import ctypes
        arch = ctypes.create_string_buffer(64)
        L.irs_hip_device_arch(0, arch, 64)
        dev = "cpu" if arch.value.endswith(b"-sim") else "cuda"
        lists = []
        for i, b in enumerate(batches):
            h = torch.zeros((len(filters), k), dtype=torch.int64, device=dev)
            c = torch.zeros((len(filters),), dtype=torch.int32, device=dev)
            b.results_to_device(h.data_ptr(), c.data_ptr())
            if dev == "cuda":
                torch.cuda.synchronize()
            lists.append((i, h, c))
        oh, os_, oc = distributed.gather_merge(L, 0, lists, n_segs, 0, 1, len(filters), k, dev)
        if dev == "cuda":
            torch.cuda.synchronize()
        gh = distributed.hits_from_int64(oh)
        gs, gc = os_.cpu().numpy(), oc.cpu().numpy()
        for q, rows in enumerate(merged):
            assert gc[q] == len(rows)
            got = [(float(gh[q, i]["score"]), int(gs[q, i]), int(gh[q, i]["doc"]))
                   for i in range(len(rows))]
            assert got == [(float(np.float32(a)), s, d) for a, s, d in rows], q
    for b in batches:
        b.close()
    for r in readers:
        r.close()


def case_multi_segment_batch(L, sizes=(30_000, 9_000, 140_000), max_rank=256, k=100):
    """irs_hip_batch_create_multi: ONE batch over segments of different sizes (so the shorter
    ones have empty chunk ids) gives, per segment, bit for bit what a batch on that segment
    alone gives — for every op, with a term missing from one segment, for every tile size —
    and it matches the oracle."""
    first = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    segs = [synth.build_segment(int(n), max_rank, first_doc=int(f)) for n, f in zip(sizes, first)]
    segs[1].metas[max_rank - 1]["docs_count"] = 0
    readers = [search.SegmentReader.from_synth(s, L=L) for s in segs]
    filters = standard_filters(max_rank, n_or8=3)
    for scorer in (BM25(), TFIDF(True)):
        prep = search.prepare(filters, scorer, [parity.segment_stats(s) for s in segs])
        for tile in (0, 4096, 12288):
            mb = search.QueryBatch(readers, prep, k)
            if tile:
                mb.configure(tile, 5, 0)
            mh, mc, mt = mb.run().results()
            assert mh.shape == (len(segs), len(filters), k)
            for i, (s, r) in enumerate(zip(segs, readers)):
                b = r.batch(prep, k)
                if tile:
                    b.configure(tile, 5, 0)
                h, c, t = b.run().results()
                b.close()
                assert np.array_equal(mc[i], c) and np.array_equal(mt[i], t), (i, tile)
                assert np.array_equal(mh[i], h), (i, tile)
                if tile == 0:
                    parity.check_single_segment(s, filters, scorer, k, mh[i], mc[i], mt[i], segs)
            mb.close()
    for r in readers:
        r.close()


def case_shared_threshold(L, sizes=(70_000, 30_000, 140_000, 50_000), max_rank=256):
    """irs_hip_batch_set_shared_threshold: the units of a query on the batch's segments share one
    threshold.  The MERGED top k is bit for bit what the batch gives without the option (and what
    the oracle's heap over all segments holds), total hits are unchanged, a segment's list is a
    prefix of its own top k — and shorter wherever the segment holds fewer of the k best docs."""
    first = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    segs = [synth.build_segment(int(n), max_rank, first_doc=int(f)) for n, f in zip(sizes, first)]
    segs[1].metas[max_rank - 1]["docs_count"] = 0   # a term missing from one segment
    readers = [search.SegmentReader.from_synth(s, L=L) for s in segs]
    ranks = synth.make_queries(6, 8, 12, max_rank, synth.SEED + 9)
    filters = [Or([by_term(int(r) - 1) for r in row]) for row in ranks]
    filters += [by_term(max_rank - 1), Or([by_term(20), by_term(max_rank - 1, 2.0)]),
                Or([by_term(12), by_term(13), by_term(14)], min_match=2),
                And([by_term(12), by_term(13)])]
    # (a TF-IDF score bound depends on the segment's largest frequencies: its units form groups
    # only where those agree)
    for scorer, grouped in ((BM25(), True), (BM25(1.2, 0.0), True), (TFIDF(True), False)):
        prep = search.prepare(filters, scorer, [parity.segment_stats(s) for s in segs])
        for k in (10, 300):
            plain = search.QueryBatch(readers, prep, k).set_path(_lib.PATH_JOINED)
            ph, pc, pt = plain.run().results()
            shared = search.QueryBatch(readers, prep, k).set_shared_threshold(True)
            sh, sc, st = shared.set_path(_lib.PATH_JOINED).run().results()
            assert shared.reruns() == 0
            assert np.array_equal(pt, st)
            assert np.all(sc <= pc)
            for i in range(len(segs)):
                for q in range(len(filters)):
                    n = int(sc[i, q])
                    assert np.array_equal(sh[i, q, :n], ph[i, q, :n]), (i, q)
            if grouped and k == 300:   # the segments share the work of finding k docs
                assert int(sc[:, :6].sum()) < int(pc[:, :6].sum())
            mp = search.merge_topk_host([(ph[i], pc[i]) for i in range(len(segs))], k)
            ms = search.merge_topk_host([(sh[i], sc[i]) for i in range(len(segs))], k)
            assert mp == ms
            ref = parity.oracle_topk(segs, filters, scorer, k)
            for q, (rows, (ohits, total)) in enumerate(zip(ms, ref)):
                assert len(rows) == len(ohits), q
                if rows:
                    a = np.array([r[0] for r in rows], np.float32)
                    assert np.allclose(a, np.sort(ohits["score"])[::-1], rtol=parity.REL_TOL, atol=0), q
            plain.close()
            shared.close()
    for r in readers:
        r.close()


def case_shared_threshold_misled(L, k=400):
    """... and when the pilot sample misleads the shared threshold (the sampled tiles of every
    segment hold far better docs than the rest) the group check behind k_select notices that the
    segments together returned fewer than k docs although more matched: one re-run with the
    sound threshold, as for a single segment (case_pilot_misled)."""
    tile, stride, n_tiles = 12288, 16, 64
    n_docs = tile * n_tiles
    segs, readers = [], []
    for si in range(2):
        rng = np.random.default_rng(31 + si)
        phase = ((si * 3 + 0) * 7) % stride     # of unit (segment si, query 0): k_join_pilot
        sampled = [t for t in range(n_tiles) if t % stride == phase]
        hot = np.concatenate([1 + t * tile + np.sort(rng.choice(tile, 400, replace=False))
                              for t in sampled]).astype(np.uint32)
        hot_f = rng.integers(1, 64, hot.size).astype(np.uint32)
        cold = np.sort(rng.choice(n_docs, 3000, replace=False)).astype(np.uint32) + 1
        seg, sr = open_lists(L, [(hot, hot_f), (cold, np.ones(cold.size, np.uint32))], n_docs,
                             synth.LAYOUT_SIMD4, norms=False)
        segs.append(seg)
        readers.append(sr)
    filters = [by_term(0), Or([by_term(0), by_term(1)]), by_term(1)]
    scorer = BM25(1.2, 0.0)
    prep = search.prepare(filters, scorer, [parity.segment_stats(s) for s in segs])
    b = search.QueryBatch(readers, prep, k).configure(0, stride, 0).set_shared_threshold(True)
    assert b.reruns() == 0
    h, c, t = b.run().results()
    assert b.reruns() == 1 and b.path() == _lib.PATH_JOINED
    merged = search.merge_topk_host([(h[i], c[i]) for i in range(2)], k)
    ref = parity.oracle_topk(segs, filters, scorer, k)
    for q, (rows, (ohits, total)) in enumerate(zip(merged, ref)):
        assert len(rows) == len(ohits), q
        a = np.array([r[0] for r in rows], np.float32)
        assert np.allclose(a, np.sort(ohits["score"])[::-1], rtol=parity.REL_TOL, atol=0), q
    b.close()
    for r in readers:
        r.close()


def case_merge_ties(L, n_lists=5, nq=7, k=64, seed=11):
    """irs_hip_merge_topk alone: few distinct scores (heavy ties across segments), ragged
    counts including empty lists, segment ids not in list order.  Expected order:
    (score desc, segment asc, doc asc) — tests/search/wand_test.cpp:72-86."""
    import ctypes
    import torch
    from iresearch_amd import _lib, distributed
    arch = ctypes.create_string_buffer(64)
    L.irs_hip_device_arch(0, arch, 64)
    dev = "cpu" if arch.value.endswith(b"-sim") else "cuda"
    rng = np.random.default_rng(seed)
    seg_ids = rng.permutation(n_lists).astype(np.uint32)
    levels = np.array([0.5, 1.25, 1.25000012, 3.0, 7.5], np.float32)
    hs, cs, expect = [], [], [[] for _ in range(nq)]
    for l in range(n_lists):
        h = np.zeros((nq, k), _lib.HIT)
        c = np.zeros(nq, np.uint32)
        for q in range(nq):
            n = int(rng.integers(0, k + 1)) if (q + l) % 4 else (0 if q % 2 else k)
            sc = np.sort(rng.choice(levels, n))[::-1]
            docs = np.zeros(n, np.uint32)
            # doc ascending inside a run of equal scores (what k_select emits)
            for v in np.unique(sc):
                m = sc == v
                docs[m] = np.sort(rng.choice(100000, int(m.sum()), replace=False)) + 1
            h[q, :n]["score"], h[q, :n]["doc"], c[q] = sc, docs, n
            expect[q] += [(float(a), int(seg_ids[l]), int(d)) for a, d in zip(sc, docs)]
        hs.append(torch.from_numpy(h.view(np.int64).reshape(nq, k).copy()).to(dev))
        cs.append(torch.from_numpy(c.view(np.int32).copy()).to(dev))
    out_h = torch.zeros((nq, k), dtype=torch.int64, device=dev)
    out_s = torch.zeros((nq, k), dtype=torch.int32, device=dev)
    out_c = torch.zeros((nq,), dtype=torch.int32, device=dev)
    lists = (ctypes.c_void_p * n_lists)(*[t.data_ptr() for t in hs])
    counts = (ctypes.c_void_p * n_lists)(*[t.data_ptr() for t in cs])
    _lib.check(L, L.irs_hip_merge_topk(0, lists, counts, seg_ids.ctypes.data, n_lists, nq, k,
                                       out_h.data_ptr(), out_s.data_ptr(), out_c.data_ptr(),
                                       None), "merge")
    if dev == "cuda":
        torch.cuda.synchronize()
    gh = distributed.hits_from_int64(out_h)
    gs, gc = out_s.cpu().numpy(), out_c.cpu().numpy()
    for q in range(nq):
        want = sorted(expect[q], key=lambda t: (-t[0], t[1], t[2]))[:k]
        assert gc[q] == len(want), q
        got = [(float(gh[q, i]["score"]), int(gs[q, i]), int(gh[q, i]["doc"]))
               for i in range(len(want))]
        assert got == want, q


# ------------------------------------------------------------------ errors --

# --------------------------------------------------------------- positions --

def _random_pos_list(rng, n, num_docs, max_tf, max_pos=400):
    d = np.sort(rng.choice(np.arange(1, num_docs + 1), n, replace=False)).astype(np.uint32)
    f = rng.integers(1, max_tf + 1, n).astype(np.uint32)
    p = np.concatenate([np.sort(rng.choice(np.arange(1, max_pos), int(k), replace=False))
                        for k in f]).astype(np.uint32)
    return d, f, p


def case_decode_positions(L, layout):
    """irs_hip_decode_positions == the emitter's input == the oracle's position iterator
    (position::next / refill / read_tail_block, formats_10.cpp:1515-1633): synthetic corpus
    plus explicit lists hitting every framing case of the `.pos` stream."""
    seg = synth.build_segment(20_000, 96, keep_postings=True, with_positions=True, layout=layout)
    ver = C.c_int32()
    assert oracle.lib().orc_check_pos_header(seg.pos_file.ctypes.data, seg.pos_file.size,
                                             C.byref(ver)) > 0
    sr = search.SegmentReader.from_synth(seg, L=L)
    for r in (1, 2, 3, 9, 40, 95, 96):
        got = sr.decode_positions(r - 1)
        assert np.array_equal(got, seg.positions[r]), ("positions", layout, r)
        assert np.array_equal(got, oracle.decode_positions(seg.doc_file, seg.pos_file,
                                                           seg.meta(r), layout)), ("oracle", r)
        # the skip() path of the reference (docs not drained) agrees on the drained ones
        part = oracle.decode_positions(seg.doc_file, seg.pos_file, seg.meta(r), layout, stride=5)
        keep = part != 0
        assert np.array_equal(got[keep], part[keep])
    sr.close()

    rng = np.random.default_rng(5)
    N = 3000
    lists = [
        _random_pos_list(rng, 1, N, 1),        # single doc, one position (nothing but a vint)
        _random_pos_list(rng, 1, N, 300),      # single doc, possibly > 128 positions
        _random_pos_list(rng, 64, N, 2),       # < 128 positions in all: tail only
        _random_pos_list(rng, 128, N, 1),      # exactly one pos block, tail_start invalid
        _random_pos_list(rng, 129, N, 3),
        _random_pos_list(rng, 700, N, 6),
        _random_pos_list(rng, 2500, N, 4),
    ]
    # total frequency exactly 128 over 64 docs; constant deltas -> ALL_EQUAL pos blocks
    d = np.arange(1, 65, dtype=np.uint32)
    lists.append((d, np.full(64, 2, np.uint32), np.tile(np.array([3, 6], np.uint32), 64)))
    d = np.arange(1, 601, dtype=np.uint32)
    lists.append((d, np.full(600, 3, np.uint32), np.tile(np.array([7, 14, 21], np.uint32), 600)))
    # wide values: positions up to 2^31 (32-bit packed blocks)
    d, f, _ = _random_pos_list(rng, 300, N, 2)
    p = np.concatenate([np.cumsum(rng.integers(1, 2**30, int(k))) for k in f]).astype(np.uint32)
    lists.append((d, f, p))
    seg = synth.segment_from_lists(lists, N, layout)
    sr = search.SegmentReader.from_synth(seg, L=L)
    for t, (d, f, p) in enumerate(lists):
        assert_decode(sr, seg, t, d, f)
        got = sr.decode_positions(t)
        assert np.array_equal(got, p), ("explicit positions", layout, t)
        assert np.array_equal(got, oracle.decode_positions(seg.doc_file, seg.pos_file,
                                                           seg.metas[t], layout)), t
    # skip data of a POS field: pend_pos / pos_ptr of every level-0 entry (ReadState
    # :1063-1080) are consistent with the frequencies
    last, ptrs, pend, pptr = oracle.read_skip0_pos(seg.doc_file, seg.metas[6])
    cum = np.cumsum(lists[6][1])
    assert np.array_equal(pend, cum[127::128][:len(pend)] % 128)
    assert (np.diff(pptr.astype(np.int64)) >= 0).all()
    last_gpu, _ = sr.term_directory(6)
    assert np.array_equal(last_gpu[:len(last)], last)
    sr.close()

    # formats 1_0 / 1_2simd (PostingsFormat 0 / 1): one-based position storage — the first delta
    # of a doc is relative to pos_limits::min() and the reader adds it back (:1589-1591, 1623-1625)
    oseg = synth.segment_from_lists(lists, N, layout, one_based=True)
    assert oseg.doc_file[4 + 1 + 31 + 3] == layout    # version 0 (scalar) / 1 (simd4)
    osr = search.SegmentReader.from_synth(oseg, L=L)
    for t, (d, f, p) in enumerate(lists):
        assert np.array_equal(osr.decode_positions(t), p), ("one-based positions", layout, t)
        assert np.array_equal(p, oracle.decode_positions(oseg.doc_file, oseg.pos_file, oseg.metas[t],
                                                         layout, one_based=True))
        zp = oracle.decode_positions(oseg.doc_file, oseg.pos_file, oseg.metas[t], layout)
        assert not np.array_equal(zp, p)             # read as zero-based it is off by one
    osr.close()

    # a POS field indexed WITH a scorer: wand data sits in front of short tails and in every
    # skip entry next to the POS fields (formats_10.cpp:511-518, 990-999)
    norms = np.full(N, 255, np.uint8)
    wseg = synth.segment_from_lists(lists[2:7], N, layout, norms, (synth.WAND_MIN_NORM,))
    assert wseg.wand_count == 1 and wseg.doc_file.size > 0
    wsr = search.SegmentReader.from_synth(wseg, L=L)
    for t, (d, f, p) in enumerate(lists[2:7]):
        dd, ff = wsr.decode_term(t)
        assert np.array_equal(dd, d) and np.array_equal(ff, f)
        assert np.array_equal(wsr.decode_positions(t), p), ("wand + positions", layout, t)
        assert np.array_equal(p, oracle.decode_positions(wseg.doc_file, wseg.pos_file,
                                                         wseg.metas[t], layout, wand_count=1))
    l2, _, pend2, _ = oracle.read_skip0_pos(wseg.doc_file, wseg.metas[4], wand_count=1)
    assert np.array_equal(l2, last) and np.array_equal(pend2, pend)
    wsr.close()


def run_phrases(L, seg, phrases, scorer, k, sr=None, cap=0, stride=0):
    own = sr is None
    sr = sr or search.SegmentReader.from_synth(seg, L=L)
    prep = search.prepare(phrases, scorer, [parity.segment_stats(seg)])
    b = sr.batch(prep, k)
    if cap or stride:
        b.configure(0, stride, cap)
    hits, counts, totals = b.run().results()
    parity.check_phrase_segment(seg, phrases, scorer, k, hits, counts, totals)
    reruns = b.reruns()
    b.close()
    if own:
        sr.close()
    return hits, counts, totals, reruns


def case_phrase_queries(L, layout, num_docs=30_000):
    """by_phrase with fixed offsets (FixedPhraseQuery::execute, phrase_query.cpp:44-111;
    PhraseIterator + FixedPhraseFrequency, phrase_iterator.hpp:75-166, 540-626) against
    the oracle's restatement: exact hit counts and phrase frequencies, scores <= 1e-5."""
    seg = synth.build_segment(num_docs, 128, with_positions=True, layout=layout)
    sr = search.SegmentReader.from_synth(seg, L=L)
    phrases = [
        by_phrase([0, 1]), by_phrase([2, 0]), by_phrase([0, 0]),         # "a a": one term twice
        by_phrase([1, 4, 0]), by_phrase([0, 3], [0, 3]),                  # a gap of two words
        by_phrase([9, 19]), by_phrase([0, 1, 2, 3]), by_phrase([5]),      # one word
        by_phrase([0, 1, 0, 2, 1, 0, 3, 0], [0, 1, 2, 4, 5, 7, 8, 9]),    # the longest allowed
        by_phrase([120, 127]),                                             # (almost) no match
        by_phrase([0, 1], boost=2.5), by_phrase([1, 0, 1], [0, 2, 4]),
        by_phrase([3, 10_000]),                                            # a term absent here
    ]
    for scorer in (BM25(), BM25(1.2, 0.0), BM25(0.0, 0.0), TFIDF(False), TFIDF(True)):
        for k in (10, 1000):
            run_phrases(L, seg, phrases, scorer, k, sr=sr)
    # a pilot pass over every 2nd / every lead block picks a threshold bin: frequent phrases
    # then only append the matches that can still make the top k — same results
    h0, c0, t0, _ = run_phrases(L, seg, phrases, BM25(), 10, sr=sr)
    for stride in (2, 1):
        h1, c1, t1, _ = run_phrases(L, seg, phrases, BM25(), 10, sr=sr, stride=stride)
        assert np.array_equal(h0, h1) and np.array_equal(c0, c1) and np.array_equal(t0, t1), stride
    # more matches than candidate slots: the buffer grows and the batch is re-run, exact
    _, _, totals, reruns = run_phrases(L, seg, phrases[:3], BM25(), 16, sr=sr, cap=64)
    assert reruns >= 1 and int(totals.max()) > 64
    sr.close()


def case_scored_expansion(L, layout=synth.LAYOUT_SIMD4, sizes=(60_000, 25_000), max_rank=512):
    """by_prefix / by_wildcard / by_range WITH scorers — what the reference harness builds for its
    Prefix3 / Wildcard tasks (index-search.cpp:363-399: scored_terms_limit): the visited terms'
    `limit` longest (segment, term) states scored as a disjunction, the rest as one unscored
    bitset (limited_sample_collector.hpp, multiterm_query.cpp:112-184).  One and two segments,
    limits 16 / 3 / 1 / 0, k below and above the number of docs that score, a visit that exists
    in one segment only, deleted documents."""
    segs = [synth.build_segment(int(n), max_rank, layout=layout, first_doc=int(f))
            for n, f in zip(sizes, np.concatenate([[0], np.cumsum(sizes)[:-1]]))]
    segs[-1].metas[max_rank - 2]["docs_count"] = 0          # a term the last segment does not hold
    rng = np.random.default_rng(5)
    for n_segs in (1, len(segs)):
        use = segs[:n_segs]
        if n_segs > 1:
            use[0].doc_mask = (rng.choice(use[0].num_docs, use[0].num_docs // 30, replace=False) + 1).astype(np.uint32)
        readers = [search.SegmentReader.from_synth(x, L=L) for x in use]
        stats = [parity.segment_stats(x) for x in use]
        ranges = [(0, 40), (300, 340), (100, 101), (max_rank - 24, max_rank), (7, 8), (200, 264)]
        visits = []
        for lo, hi in ranges:
            per_seg = []
            for x in use:     # the visit: the terms of the range the segment holds, in term order
                per_seg.append(np.array([t for t in range(lo, hi) if int(x.metas[t]["docs_count"])], np.uint32))
            visits.append(per_seg)
        visits.append([np.arange(0, 5, dtype=np.uint32)] + [np.zeros(0, np.uint32)] * (n_segs - 1))
        for scorer in (BM25(), TFIDF(False)):
            for limit in (16, 3, 1, 0):
                for k in (10, 3000):
                    prep = search.prepare_expansions(visits, limit, scorer, stats)
                    assert all(len(p.scored) <= max(limit, 0) for p in prep)
                    h, c, t = search.execute_expansions(readers, prep, k)
                    parity.check_expansions(use, visits, limit, scorer, k, h, c, t)
        for r in readers:
            r.close()


def case_conj_sparse_lead(L, layout=synth.LAYOUT_SIMD4, n_docs=400_000):
    """Conjunctions whose rarest term is FAR rarer than the others (the reference's AndHighLow
    class): a lead block's 128 docs then fall into up to 128 different blocks of every other term,
    and the block is cut into pieces, a wavefront each (ConjItem: build_conj_work's rule; the
    pieces find their own start in the other terms' directories).  Explicit lists: leads of two
    blocks + a tail, of a tail only and of one doc against dense terms; with and without block-max
    pruning, deleted docs, and every forced cut (IRS_HIP_CONJ_SPLIT_LOG2) — the same results,
    bit for bit, as uncut."""
    import os
    rng = np.random.default_rng(77)
    pick = lambda n: np.sort(rng.choice(n_docs, n, replace=False)).astype(np.uint32) + 1
    lists = [np.arange(2, n_docs, 3, dtype=np.uint32),      # 0: dense, regular
             pick(300),                                      # 1: two blocks + a tail of 44
             pick(n_docs // 2),                              # 2: dense, random
             pick(40),                                       # 3: tail only
             np.array([n_docs // 2 + 1], np.uint32),         # 4: one doc
             pick(128 * 5)]                                  # 5: exactly five blocks, no tail
    lists[1][:3] = [2, 5, 8]                                 # (some docs of the dense term for sure)
    lists[1] = np.unique(lists[1])
    freqs = [rng.integers(1, 6, l.size).astype(np.uint32) for l in lists]
    norms = rng.integers(20, 200, n_docs).astype(np.uint8)
    seg = synth.segment_from_lists(list(zip(lists, freqs)), n_docs, layout, norms=norms)
    st = [parity.segment_stats(seg)]
    filters = [And([by_term(1), by_term(0)]), And([by_term(1), by_term(0), by_term(2)]),
               And([by_term(3), by_term(0)]), And([by_term(4), by_term(0)]), And([by_term(4), by_term(2)]),
               And([by_term(3), by_term(2), by_term(0)]), And([by_term(1), by_term(2)]),
               And([by_term(5), by_term(2)]), And([by_term(5), by_term(0), by_term(2)]),
               And([by_term(0), by_term(2)]), And([by_term(1), by_term(5)])]
    for masked in (False, True):
        if masked:
            seg.doc_mask = np.concatenate([lists[1][::7], lists[5][::3], pick(n_docs // 10)])
        sr = search.SegmentReader.from_synth(seg, L=L)
        for scorer in (BM25(), TFIDF(True)):
            ref = None
            for forced in (None, "0", "1", "2", "3", "4"):
                if forced is None:
                    os.environ.pop("IRS_HIP_CONJ_SPLIT_LOG2", None)
                else:
                    os.environ["IRS_HIP_CONJ_SPLIT_LOG2"] = forced
                try:
                    for wand in (False, True):
                        b = sr.batch(search.prepare(filters, scorer, st), 50).set_path(_lib.PATH_ITEMS)
                        if wand:
                            b.set_wand(True)
                        h, c, t = b.run().results()
                        b.close()
                        if ref is None:
                            parity.check_single_segment(seg, filters, scorer, 50, h, c, t)
                            ref = (h.copy(), c.copy(), t.copy())
                            assert int(t[0]) > 0 and int(t[3]) <= 1
                        assert np.array_equal(h, ref[0]) and np.array_equal(c, ref[1]), (masked, forced, wand)
                        if not wand:   # (pruned runs only count the docs they evaluated)
                            assert np.array_equal(t, ref[2]), (masked, forced)
                finally:
                    os.environ.pop("IRS_HIP_CONJ_SPLIT_LOG2", None)
        sr.close()


def case_doc_mask(L, layout=synth.LAYOUT_SIMD4, num_docs=70_000, max_rank=256):
    """A segment with deleted documents (irs_hip_segment_desc.doc_mask — the DocumentMask the
    reference wraps every iterator with: SegmentReaderImpl::mask -> MaskDocIterator,
    core/index/segment_reader_impl.cpp:69-101, 286): 5 % random deletions plus the first doc, the
    last doc, a run longer than a posting block and every doc of one term.  Totals, doc sets, scores
    and order equal the oracle's masked run on every execution path (work items, joined streams,
    block-driven conjunctions with and without block-max pruning, phrases, 64-bit accumulators);
    bit_union leaves the deleted docs out; the postings-level surfaces are not filtered."""
    import os
    seg = synth.build_segment(num_docs, max_rank, layout=layout, with_positions=True)
    rng = np.random.default_rng(2026)
    victim = max_rank - 3                                   # a rare term: all of its docs go
    vd, _ = oracle.decode_term(seg.doc_file, seg.metas[victim], layout)
    mask = np.concatenate([
        rng.choice(num_docs, num_docs // 20, replace=False).astype(np.uint32) + 1,
        np.array([1, num_docs, 0, num_docs + 5, 7, 7], np.uint32),     # ends, out of range, twice
        np.arange(20_000, 20_400, dtype=np.uint32), vd.astype(np.uint32)])
    rng.shuffle(mask)
    gone = np.unique(mask[(mask >= 1) & (mask <= num_docs)])
    plain = search.SegmentReader.from_synth(seg, L=L)        # the same segment without its mask
    seg.doc_mask = mask
    sr = search.SegmentReader.from_synth(seg, L=L)
    assert sr.live_docs_count() == num_docs - gone.size and plain.live_docs_count() == num_docs
    st = [parity.segment_stats(seg)]
    ranks = synth.make_queries(5, 8, 2, max_rank, synth.SEED + 9)
    filters = [Or([by_term(int(r) - 1) for r in row]) for row in ranks]
    filters += standard_filters(max_rank, n_or8=1)
    filters += [by_term(victim), And([by_term(victim), by_term(0)]), Or([by_term(victim), by_term(victim - 1)]),
                And([by_term(max_rank - 1), by_term(0)]), And([by_term(0), by_term(1)]),
                And([by_term(0), by_term(1), by_term(2), by_term(3), by_term(4)])]
    for scorer in (BM25(), TFIDF(True), TFIDF(False)):
        for path in (_lib.PATH_ITEMS, _lib.PATH_JOINED, _lib.PATH_AUTO):
            for k in (25, 1000):
                b = sr.batch(search.prepare(filters, scorer, st), k).set_path(path)
                h, c, t = b.run().results()
                parity.check_single_segment(seg, filters, scorer, k, h, c, t)
                b.close()
        # (the victim's docs are all gone)
        assert int(t[len(filters) - 6]) == 0 and int(t[len(filters) - 5]) == 0
    # ... the mask really changed something: the unmasked reader counts more
    b = plain.batch(search.prepare(filters[:5], BM25(), st), 25)
    _, _, t_plain = b.run().results()
    b.close()
    b = sr.batch(search.prepare(filters[:5], BM25(), st), 25)
    _, _, t_masked = b.run().results()
    b.close()
    assert (t_masked < t_plain).all()
    # block-max pruning (ExecutionContext::wand) and 64-bit accumulators
    b = sr.batch(search.prepare(filters, BM25(), st), 25).set_wand(True)
    h, c, t = b.run().results()
    b.close()
    b = sr.batch(search.prepare(filters, BM25(), st), 25)
    h0, c0, _ = b.run().results()
    b.close()
    assert np.array_equal(h, h0) and np.array_equal(c, c0)     # wand == exhaustive, masked too
    os.environ["IRS_HIP_ACC"] = "64"
    try:
        b = sr.batch(search.prepare(filters, BM25(), st), 25)
        h, c, t = b.run().results()
        parity.check_single_segment(seg, filters, BM25(), 25, h, c, t)
        b.close()
    finally:
        del os.environ["IRS_HIP_ACC"]
    # by_phrase
    phrases = [by_phrase([0, 1]), by_phrase([2, 0]), by_phrase([1, 4, 0]), by_phrase([5]),
               by_phrase([victim]), by_phrase([max_rank - 1, 0]), by_phrase([0, 3], [0, 3])]
    for scorer in (BM25(), TFIDF(False)):
        for k in (10, 1000):
            run_phrases(L, seg, phrases, scorer, k, sr=sr)
    # bit_union: what segment.mask(lazy_bitset_iterator) yields
    n_words = (num_docs + 64) // 64
    terms = [0, 5, victim, max_rank - 1]
    got, cnt = sr.bit_union(terms, n_words)
    want, ocnt = oracle.bit_union(seg.doc_file, [seg.metas[x] for x in terms], layout, True, n_words)
    dead = np.zeros(n_words, np.uint64)
    np.bitwise_or.at(dead, gone // 64, np.uint64(1) << (gone % 64).astype(np.uint64))
    assert cnt == ocnt and np.array_equal(got, want & ~dead) and not np.array_equal(got, want)
    assert int(sr.bit_union_counts([terms])[0]) == int(np.unpackbits((want & ~dead).view(np.uint8)).sum())
    # postings-level surfaces are the postings_reader's: not filtered
    d, f = sr.decode_term(victim)
    assert np.array_equal(d, vd)
    sr.close()
    plain.close()


def case_phrase_ragged(L, layout=synth.LAYOUT_SIMD4, one_based=False):
    """Explicit lists: single-doc terms, lists shorter than a block, phrases that exist
    only across block / tile borders, very frequent terms in one doc."""
    N = 9000
    rng = np.random.default_rng(3)
    a = _random_pos_list(rng, 4000, N, 5, 60)
    b = _random_pos_list(rng, 3500, N, 5, 60)
    c = _random_pos_list(rng, 90, N, 3, 60)
    one = (np.array([2049], np.uint32), np.array([3], np.uint32), np.array([5, 6, 9], np.uint32))
    # a doc in which both terms occur 200 times, interleaved: "x y" matches 200 times
    dd = np.array([7, 2048, 2049, 8999], np.uint32)
    x = (dd, np.full(4, 200, np.uint32), np.tile(np.arange(1, 401, 2, dtype=np.uint32), 4))
    y = (dd, np.full(4, 200, np.uint32), np.tile(np.arange(2, 402, 2, dtype=np.uint32), 4))
    # ... and docs in which they occur 500 / 150 times: more positions than the wavefront's merge
    # stages at once (384 of the first term: one lane walks that doc; 3 x 150: two passes)
    dz = np.array([100, 101, 102, 103, 5000], np.uint32)
    fz = np.array([150, 150, 150, 500, 500], np.uint32)
    z = (dz, fz, np.concatenate([np.arange(1, 2 * n, 2, dtype=np.uint32) for n in fz]))
    w = (dz, fz, np.concatenate([np.arange(2, 2 * n + 1, 2, dtype=np.uint32) for n in fz]))
    lists = [a, b, c, one, x, y, z, w]
    seg = synth.segment_from_lists(lists, N, layout, norms=np.full(N, 255, np.uint8),
                                   one_based=one_based)
    phrases = [by_phrase([0, 1]), by_phrase([1, 0]), by_phrase([0, 2]), by_phrase([2, 1, 0]),
               by_phrase([3, 3]), by_phrase([0, 3]), by_phrase([4, 5]), by_phrase([5, 4]),
               by_phrase([4, 5, 4]), by_phrase([4, 4], [0, 2]), by_phrase([3, 4], [0, 2]),
               by_phrase([6, 7]), by_phrase([7, 6]), by_phrase([6, 7, 6]), by_phrase([6, 6], [0, 2]),
               by_phrase([4, 7])]
    for scorer in (BM25(), TFIDF(True)):
        for k in (3, 100):
            run_phrases(L, seg, phrases, scorer, k)


def case_phrase_fuzz(L, iters=12, seed=1):
    """Seeded differential run: random small segments (single-doc terms, short and long
    lists, both layouts), random phrases of 1..5 terms — repeated terms, gaps, equal offsets —
    random k and scorer, against the oracle's phrase iterator."""
    rng = np.random.default_rng(seed)
    for it in range(iters):
        N = int(rng.integers(50, 6000))
        nterms = int(rng.integers(2, 7))
        lists = []
        for _ in range(nterms):
            n = 1 if rng.random() < 0.15 else int(rng.integers(1, min(N, 1500)))
            maxtf = int(rng.integers(1, 9))
            lists.append(_random_pos_list(rng, n, N, maxtf, int(rng.integers(maxtf + 2, 40))))
        seg = synth.segment_from_lists(lists, N, int(rng.integers(0, 2)),
                                       norms=rng.integers(40, 255, N).astype(np.uint8))
        phrases = []
        for _ in range(8):
            m = int(rng.integers(1, 6))
            offs = [0]
            for _ in range(m - 1):
                offs.append(offs[-1] + int(rng.integers(0, 4)))
            phrases.append(by_phrase([int(rng.integers(0, nterms)) for _ in range(m)], offs))
        scorer = [BM25(), TFIDF(True), BM25(1.2, 0.0)][it % 3]
        run_phrases(L, seg, phrases, scorer, int(rng.integers(1, 50)))


def case_phrase_multi_segment(L, sizes=(12_000, 5_000, 30_000), k=50):
    """One batch over several segments of one device (irs_hip_batch_create_multi): every
    segment's phrase results equal its own run and the oracle's multi-segment harness."""
    segs, first = [], 0
    for n in sizes:
        segs.append(synth.build_segment(n, 64, first_doc=first, with_positions=True))
        first += n
    srs = [search.SegmentReader.from_synth(s, L=L) for s in segs]
    phrases = [by_phrase([0, 1]), by_phrase([2, 1, 0]), by_phrase([7, 3]), by_phrase([0, 5], [0, 2])]
    stats = [parity.segment_stats(s) for s in segs]
    scorer = BM25()
    prep = search.prepare(phrases, scorer, stats)
    b = search.QueryBatch(srs, prep, k)
    hits, counts, totals = b.run().results()
    for i, s in enumerate(segs):
        parity.check_phrase_segment(s, phrases, scorer, k, hits[i], counts[i], totals[i], segs)
    merged = search.merge_topk_host([(hits[i], counts[i]) for i in range(len(segs))], k)
    views = [parity.oracle_view(s) for s in segs]
    osc = parity.oracle_scorer(scorer)
    for q, ph in enumerate(phrases):
        metas = np.stack([parity.metas_for(s, ph.terms) for s in segs])
        oh, total = oracle.search_phrase(views, metas, ph.offsets, osc, k, ph.boost)
        assert total == int(totals[:, q].sum())
        assert len(merged[q]) == len(oh)
        got = np.array([m[0] for m in merged[q]], np.float32)
        want = np.sort(oh["score"])[::-1]
        assert np.allclose(got, want, rtol=parity.REL_TOL, atol=0), ("merged scores", q)
    b.close()
    for sr in srs:
        sr.close()


def phrase_golden_corpus():
    """tests/golden/phrase_golden.json: the corpus tests/resources/phrase_sequential.json
    and the doc sets tests/search/phrase_filter_tests.cpp asserts for its plain-term
    phrases.  Field `phrase_anl`: text analyzer, locale C, no stopwords -> words at
    positions 1, 2, ... (doc_generator.hpp:617-625)."""
    import json
    g = json.loads((GOLDEN / "phrase_golden.json").read_text())
    names = [d["name"] for d in g["corpus"]]
    toks = [d["phrase"].split() for d in g["corpus"]]
    vocab = sorted({w for t in toks for w in t})
    lists = []
    for w in vocab:
        d, f, p = [], [], []
        for i, t in enumerate(toks):
            pos = [j + 1 for j, x in enumerate(t) if x == w]
            if pos:
                d.append(i + 1)
                f.append(len(pos))
                p.extend(pos)
        lists.append((np.array(d, np.uint32), np.array(f, np.uint32), np.array(p, np.uint32)))
    norms = np.array([len(t) for t in toks], np.uint8)
    return names, vocab, lists, norms, g["vectors"] + [dict(v, docs=sorted(v["ranked"]))
                                                        for v in g.get("ranked", [])]


def case_phrase_reference_vectors(L, layout=synth.LAYOUT_SIMD4):
    """The reference's own phrase expectations, through the C ABI (and the oracle)."""
    names, vocab, lists, norms, vectors = phrase_golden_corpus()
    seg = synth.segment_from_lists(lists, len(names), layout, norms)
    sr = search.SegmentReader.from_synth(seg, L=L)
    view = parity.oracle_view(seg)
    assert len(vectors) >= 7
    for scorer in (BM25(), TFIDF(True)):
        phrases, expect = [], []
        for v in vectors:
            if any(w not in vocab for w in v["words"]) or "ranked" in v:
                continue
            phrases.append(by_phrase([vocab.index(w) for w in v["words"]], v["offsets"]))
            expect.append(v["docs"])
        hits, counts, totals, _ = run_phrases(L, seg, phrases, scorer, 64, sr=sr)
        osc = parity.oracle_scorer(scorer)
        for q, (ph, want) in enumerate(zip(phrases, expect)):
            got = sorted(int(d) for d in hits[q, :counts[q]]["doc"])
            assert [names[d - 1] for d in got] == want, ("golden docs", q, got, want)
            assert int(totals[q]) == len(want)
            oh, total = oracle.search_phrase([view], parity.metas_for(seg, ph.terms)[None, :],
                                             ph.offsets, osc, 64)
            assert total == len(want)
            assert [names[d - 1] for d in sorted(int(x) for x in oh["doc"])] == want
    # the score ORDER bm25_test.cpp / tfidf_test.cpp (test_phrase) assert for "jumps high"
    ranked = [v for v in vectors if "ranked" in v]
    assert len(ranked) == 2
    for v in ranked:
        scorer = BM25(1.2, 0.0) if v["scorer"] == "bm25_b0" else TFIDF(False)
        ph = [by_phrase([vocab.index(w) for w in v["words"]], v["offsets"])]
        hits, counts, _, _ = run_phrases(L, seg, ph, scorer, 64, sr=sr)
        got = [names[int(d) - 1] for d in hits[0, :counts[0]]["doc"]]
        assert got == v["ranked"], ("golden ranking", v["scorer"], got)
        oh, _ = oracle.search_phrase([view], parity.metas_for(seg, ph[0].terms)[None, :],
                                     ph[0].offsets, parity.oracle_scorer(scorer), 64)
        key = sorted(zip(-oh["score"].astype(np.float64), oh["doc"].astype(np.int64)))
        assert [names[d - 1] for _, d in key] == v["ranked"]
    sr.close()


def case_phrase_errors(L):
    seg = synth.build_segment(5000, 32, with_positions=True)
    plain = synth.build_segment(5000, 32)
    stats = [parity.segment_stats(seg)]
    sr = search.SegmentReader.from_synth(seg, L=L)
    sr_plain = search.SegmentReader.from_synth(plain, L=L)
    prep = search.prepare([by_phrase([0, 1])], BM25(), stats)
    with pytest.raises(_lib.IrsHipError) as e:       # the field has no positions
        sr_plain.batch(prep, 10)
    assert e.value.status == _lib.EUNSUPPORTED
    with pytest.raises(_lib.IrsHipError) as e:
        sr_plain.decode_positions(0)
    assert e.value.status == _lib.EINVAL
    mixed = search.prepare([by_phrase([0, 1]), by_term(3)], BM25(), stats)
    with pytest.raises(_lib.IrsHipError) as e:       # phrase and boolean queries do not mix
        sr.batch(mixed, 10)
    assert e.value.status == _lib.EUNSUPPORTED
    long = search.prepare([by_phrase(list(range(_lib.MAX_PHRASE_TERMS + 1)))], BM25(), stats)
    with pytest.raises(_lib.IrsHipError) as e:
        sr.batch(long, 10)
    assert e.value.status == _lib.EINVAL
    bad = search.prepare([by_phrase([0, 1])], BM25(), stats)
    bad[0].offsets = [1, 2]                           # not relative to the first term
    with pytest.raises(_lib.IrsHipError) as e:
        sr.batch(bad, 10)
    assert e.value.status == _lib.EINVAL

    def open_with(**kw):
        args = dict(doc_file=seg.doc_file, metas=seg.metas, num_docs=seg.num_docs,
                    layout=seg.layout, norms=seg.norms, norm_width=1,
                    docs_with_field=seg.docs_with_field, total_term_freq=seg.total_term_freq,
                    L=L, pos_file=seg.pos_file)
        args.update(kw)
        return search.SegmentReader(**args)

    pbad = seg.pos_file.copy()
    pbad[6] ^= 0xFF                                   # format name
    with pytest.raises(_lib.IrsHipError) as e:
        open_with(pos_file=pbad)
    assert e.value.status == _lib.ECORRUPT
    pbad = seg.pos_file.copy()
    pbad[int(seg.metas[0]["pos_start"])] = 99         # pos block header: 99 bits
    with pytest.raises(_lib.IrsHipError) as e:
        open_with(pos_file=pbad)
    assert e.value.status == _lib.ECORRUPT
    metas = seg.metas.copy()
    metas[0]["pos_end"] += 1                          # tail is not where the blocks end
    with pytest.raises(_lib.IrsHipError) as e:
        open_with(metas=metas)
    assert e.value.status == _lib.ECORRUPT
    # term_meta::freq disagrees with the frequencies in `.doc` (by a whole pos block, so that
    # only the totals check can tell): the position kernels would run past their buffers
    for t in (0, len(seg.metas) - 1):
        metas = seg.metas.copy()
        metas[t]["freq"] += 128
        with pytest.raises(_lib.IrsHipError) as e:
            open_with(metas=metas)
        assert e.value.status == _lib.ECORRUPT, t
    with pytest.raises(_lib.IrsHipError) as e:       # positions without frequencies
        open_with(has_freq=False)
    assert e.value.status == _lib.EINVAL
    with pytest.raises(_lib.IrsHipError) as e:       # offsets / payloads change the `.pos` tail
        open_with(pos_features=1)
    assert e.value.status == _lib.EUNSUPPORTED
    sr.close()
    sr_plain.close()


def case_errors(L):
    seg = synth.build_segment(5000, 64)
    sr = search.SegmentReader.from_synth(seg, L=L)

    def open_with(**kw):
        args = dict(doc_file=seg.doc_file, metas=seg.metas, num_docs=seg.num_docs,
                    layout=seg.layout, norms=seg.norms, norm_width=1,
                    docs_with_field=seg.docs_with_field, total_term_freq=seg.total_term_freq, L=L)
        args.update(kw)
        return search.SegmentReader(**args)

    bad = seg.doc_file.copy()
    bad[0] ^= 0xFF                                   # magic
    with pytest.raises(_lib.IrsHipError) as e:
        open_with(doc_file=bad)
    assert e.value.status == _lib.ECORRUPT
    with pytest.raises(_lib.IrsHipError) as e:       # version says simd4
        open_with(layout=synth.LAYOUT_SCALAR)
    assert e.value.status == _lib.EINVAL
    bad = seg.doc_file.copy()
    bad[int(seg.metas[0]["doc_start"])] = 77         # block header: 77 bits
    with pytest.raises(_lib.IrsHipError) as e:
        open_with(doc_file=bad)
    assert e.value.status == _lib.ECORRUPT
    with pytest.raises(_lib.IrsHipError) as e:       # doc ids run past the segment
        open_with(num_docs=seg.num_docs // 2)
    assert e.value.status in (_lib.ECORRUPT, _lib.EUNSUPPORTED)
    metas = seg.metas.copy()
    metas[3]["doc_start"] = seg.doc_file.size + 5
    with pytest.raises(_lib.IrsHipError) as e:
        open_with(metas=metas)
    assert e.value.status == _lib.ECORRUPT
    with pytest.raises(_lib.IrsHipError) as e:       # no such device
        search.SegmentReader(seg.doc_file, seg.metas, seg.num_docs, seg.layout, seg.norms, 1,
                             device=99, L=L)
    assert e.value.status == _lib.EHIP

    prep = search.prepare([by_term(1)], BM25(), [parity.segment_stats(seg)])
    for k in (0, _lib.MAX_K + 1):
        with pytest.raises(_lib.IrsHipError) as e:
            sr.batch(prep, k)
        assert e.value.status == _lib.EINVAL
    many = [Or([by_term(i) for i in range(_lib.MAX_TERMS + 1)])]
    with pytest.raises(_lib.IrsHipError) as e:
        sr.batch(search.prepare(many, BM25(), [parity.segment_stats(seg)]), 10)
    assert e.value.status == _lib.EINVAL
    # candidate buffer too small -> exact re-run (full histogram, then a grown
    # buffer), never a silently wrong top-k
    fl = [Or([by_term(0), by_term(1), by_term(2)]), by_term(5)]
    run_and_check(L, seg, fl, BM25(), 16, 4096, 1000, 16, sr=sr)
    run_and_check(L, seg, fl, BM25(0.0, 0.0), 16, 4096, 1000, 16, sr=sr)  # BM1: all scores tie
    cnt = C.c_uint32()
    assert L.irs_hip_decode_term(sr.handle, 10_000, None, None, 0, C.byref(cnt)) == _lib.EINVAL
    sr.close()
