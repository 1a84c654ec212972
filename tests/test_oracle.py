"""The oracle against the reference's golden vectors and its own compiled codec
(SURVEY.md §8c), plus internal consistency of the restated search loop."""
import math
from pathlib import Path

import numpy as np
import pytest

import oracle
import parity
from iresearch_amd import search, synth
from iresearch_amd.search import BM25, TFIDF, And, Or, by_term

GOLDEN = Path(__file__).parent / "golden"
LAYOUTS = {"scalar": oracle.LAYOUT_SCALAR, "simd4": oracle.LAYOUT_SIMD4}


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN / "codec_golden.npz")


@pytest.mark.parametrize("layout", ["scalar", "simd4"])
def test_codec_matches_reference_golden(golden, layout):
    """Packed words produced by the REFERENCE's compiled packers
    (tests/golden/make_golden.py) for the literal vector of
    tests/utils/bit_packing_tests.cpp:102-114 and for random data, bits 1..32."""
    lay = LAYOUTS[layout]
    lit = golden["literal"]
    for bits in range(1, 33):
        mask = np.uint32(0xFFFFFFFF if bits == 32 else (1 << bits) - 1)
        for name, vals in (("lit", lit & mask), ("rnd", golden["random_b%d" % bits])):
            want = golden["%s_%s_b%d" % (layout, name, bits)]
            assert np.array_equal(oracle.pack(vals, bits, lay), want), (name, bits)
            assert np.array_equal(oracle.unpack(want, bits, lay), vals), (name, bits)


def test_packed_at_random_access(golden):
    """packed::at, bit_packing_tests.cpp:162-174."""
    lit = golden["literal"]
    for bits in range(1, 33):
        mask = np.uint32(0xFFFFFFFF if bits == 32 else (1 << bits) - 1)
        words = golden["scalar_lit_b%d" % bits]
        for i in range(126):
            assert oracle.lib().orc_at_scalar(words.ctypes.data, i, bits) == int(lit[i] & mask)


def test_codec_against_live_reference():
    """Only where oracle/_ref exists (build container): the reference's own
    unpackers on fresh random data."""
    R = oracle.ref()
    if R is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(5)
    for bits in range(1, 33):
        for _ in range(4):
            v = rng.integers(0, 1 << bits, 128, dtype=np.uint64).astype(np.uint32)
            for lay, rp, ru in ((0, R.ref_pack_scalar, R.ref_unpack_scalar),
                                (1, R.ref_pack_simd4, R.ref_unpack_simd4)):
                ref = np.zeros(4 * bits, np.uint32)
                rp(v.ctypes.data, bits, ref.ctypes.data)
                assert np.array_equal(oracle.pack(v, bits, lay), ref)
                out = np.zeros(128, np.uint32)
                ru(ref.ctypes.data, bits, out.ctypes.data)
                assert np.array_equal(out, v) and np.array_equal(oracle.unpack(ref, bits, lay), v)


@pytest.mark.parametrize("layout", [0, 1])
def test_reader_roundtrip_reference_lists(layout):
    """Independent emitter -> oracle reader on the reference's own test lists
    (formats_10_tests.cpp:452-457; tests/resources/postings.txt, :775-864)."""
    p = np.loadtxt(GOLDEN / "postings_6098.txt", dtype=np.uint32)
    lists = [([1, 3, 5, 7, 79, 101, 124], [10] * 7), ([2, 7, 9, 19], [10] * 4),
             (p, np.ones(p.size, np.uint32)), ([42], [3])]
    seg = synth.segment_from_lists(lists, 10_000, layout)
    for t, (d, f) in enumerate(lists):
        od, of = oracle.decode_term(seg.doc_file, seg.metas[t], layout)
        assert np.array_equal(od, np.asarray(d, np.uint32))
        assert np.array_equal(of, np.asarray(f, np.uint32))
        od2, _ = oracle.decode_term(seg.doc_file, seg.metas[t], layout, want_freq=False)
        assert np.array_equal(od2, od)
    # 6098 = 47 full blocks + 82 tail; skip entries: one per block boundary but the last
    last, ptrs, levels = oracle.read_skip0(seg.doc_file, seg.metas[2])
    assert len(last) == 47 and levels == 2  # 1 + log8(10000 / 128), skip_list.cpp:38-41
    assert np.array_equal(last, p[127::128][:47])
    ver = __import__("ctypes").c_int32()
    assert oracle.lib().orc_check_doc_header(seg.doc_file.ctypes.data, seg.doc_file.size,
                                             ver) > 0
    assert ver.value == (5 if layout else 4)


def test_bm25_stats_formulas():
    """BM25::collect (bm25.cpp:366-410) restated twice (oracle C++, host Python)
    and cross-checked in double precision."""
    for (k, b, D, d, ttf) in [(1.2, 0.75, 10_000_000, 123_456, 1_000_123_456),
                              (2.0, 1.0, 1000, 1, 5000), (1.2, 0.0, 500, 499, 777),
                              (0.0, 0.75, 500, 10, 5000)]:
        st = oracle.bm25_stats(k, b, D, d, ttf)
        hs = BM25(k, b).collect(D, d, ttf)
        assert np.float32(st.idf) == hs.idf
        assert np.float32(st.norm_const) == hs.norm_const
        if k != 0 and b != 0:
            assert np.float32(st.norm_length) == hs.norm_length
            avgdl = ttf / D
            for n in (1, 7, 100, 255):
                exact = 1.0 / (k * (1 - b) + k * b * n / avgdl)
                assert abs(st.norm_cache[n] - exact) <= 2e-6 * exact
            assert st.norm_cache[0] == 0.0
        assert abs(float(st.idf) - math.log1p((D - d + 0.5) / (d + 0.5))) < 1e-6
    assert np.float32(oracle.lib().orc_tfidf_idf(1000, 10)) == TFIDF().collect(1000, 10, 0).idf


def test_harness_matches_exhaustive_scores():
    """The restated index-search heap (index-search.cpp:719-787) returns the k best
    of the exhaustive scores; OR-2 / OR-n / AND / single term; BM25 vs float64."""
    seg = synth.build_segment(30_000, 200, keep_postings=True)
    view = parity.oracle_view(seg)
    sc = parity.oracle_scorer(BM25())
    filters = [Or([by_term(3), by_term(50), by_term(120), by_term(7)]),
               Or([by_term(10), by_term(150)]), by_term(99),
               And([by_term(2), by_term(30), by_term(60)])]
    res = parity.oracle_topk([seg], filters, BM25(), 50)
    D, ttf = seg.docs_with_field, seg.total_term_freq
    for flt, (hits, total) in zip(filters, res):
        op, subs = search._terms_of(flt)
        terms = [s.term for s in subs]
        dwt = [int(seg.metas[t]["docs_count"]) for t in terms]
        scores, matched = oracle.score_all(view, parity.metas_for(seg, terms), op, sc, D, dwt, ttf)
        assert total == int(matched.sum())
        want = np.sort(scores[matched.astype(bool)])[::-1][:50]
        assert np.array_equal(np.sort(hits["score"])[::-1], want)
        assert (hits["score"][:-1] >= hits["score"][1:]).all()
        assert np.array_equal(scores[hits["doc"]], hits["score"])
        # float64 model of BM25 over the raw postings
        exact = np.zeros(seg.num_docs + 1)
        cnt = np.zeros(seg.num_docs + 1, np.int32)
        avgdl = ttf / D
        for t in terms:
            d, f = seg.postings[t + 1]
            idf = math.log1p((D - len(d) + 0.5) / (len(d) + 0.5))
            dl = seg.norms[d - 1].astype(np.float64)
            tf = f.astype(np.float64)
            exact[d] += idf * 2.2 * tf / (tf + 1.2 * (0.25 + 0.75 * dl / avgdl))
            cnt[d] += 1
        m = cnt == len(terms) if op == oracle.OP_AND else cnt > 0
        assert np.array_equal(m, matched.astype(bool))
        assert np.allclose(scores[m], exact[m], rtol=3e-6)


def test_multi_segment_statistics_are_global():
    """idf/avgdl come from all segments (term_filter.cpp:102-125): splitting an
    index must not change scores."""
    whole = synth.build_segment(20_000, 100)
    parts = [synth.build_segment(10_000, 100, first_doc=0),
             synth.build_segment(10_000, 100, first_doc=10_000)]
    flt = [Or([by_term(5), by_term(40), by_term(77)])]
    a = parity.oracle_topk([whole], flt, BM25(), 30)[0][0]
    b = parity.oracle_topk(parts, flt, BM25(), 30)[0][0]
    assert np.array_equal(np.sort(a["score"]), np.sort(b["score"]))
    docs_b = np.sort(b["doc"] + 10_000 * b["segment"])
    ties = len(np.unique(a["score"])) < 30
    if not ties:
        assert np.array_equal(np.sort(a["doc"]), docs_b)


def test_positions_and_phrase_oracle():
    """The restated position iterator reproduces the emitter's input (all framings),
    the phrase iterator equals the set definition of a phrase match, its score is the
    scorer at tf = phrase frequency with the terms' idf summed (float64 model), and the
    doc sets the reference's phrase tests assert come out (tests/golden/phrase_golden.json)."""
    import cases
    seg = synth.build_segment(20_000, 48, keep_postings=True, with_positions=True)
    view = parity.oracle_view(seg)
    for r in (1, 5, 48):
        for stride in (1, 4):
            got = oracle.decode_positions(seg.doc_file, seg.pos_file, seg.meta(r), seg.layout,
                                          stride=stride)
            keep = got != 0
            assert np.array_equal(got[keep], seg.positions[r][keep]) and keep.sum() > 0
            assert stride > 1 or keep.all()
    D, ttf = seg.docs_with_field, seg.total_term_freq
    sc = oracle.Scorer(oracle.SCORER_BM25, 1.2, 0.75, 0)
    for ranks, offs in (([1, 2], [0, 1]), ([3, 1, 2], [0, 1, 2]), ([1, 1], [0, 2]), ([7, 30], [0, 1])):
        per = []
        for r in ranks:
            d, f = seg.postings[r]
            o = np.concatenate([[0], np.cumsum(f)]).astype(np.int64)
            per.append({int(d[i]): set(seg.positions[r][o[i]:o[i + 1]].tolist())
                        for i in range(len(d))})
        want = {}
        for doc, p0 in per[0].items():
            if all(doc in x for x in per[1:]):
                c = sum(all((q + o) in per[i][doc] for i, o in enumerate(offs)) for q in p0)
                if c:
                    want[doc] = c
        metas = np.array([seg.meta(r) for r in ranks])
        dwt = [int(seg.meta(r)["docs_count"]) for r in ranks]
        scores, pf = oracle.score_all_phrase(view, metas, offs, sc, D, dwt, ttf)
        assert {int(d): int(pf[d]) for d in np.nonzero(pf)[0]} == want
        idf = sum(math.log1p((D - n + 0.5) / (n + 0.5)) for n in dwt)
        docs = np.array(sorted(want), np.int64)
        tf = np.array([want[int(d)] for d in docs], np.float64)
        dl = seg.norms[docs - 1].astype(np.float64)
        exact = idf * 2.2 * tf / (tf + 1.2 * (0.25 + 0.75 * dl / (ttf / D)))
        assert np.allclose(scores[docs], exact, rtol=3e-6)
    names, vocab, lists, norms, vectors = cases.phrase_golden_corpus()
    gseg = synth.segment_from_lists(lists, len(names), synth.LAYOUT_SCALAR, norms)
    gview = parity.oracle_view(gseg)
    for v in vectors:
        terms = [vocab.index(w) for w in v["words"]]
        hits, total = oracle.search_phrase([gview], parity.metas_for(gseg, terms)[None, :],
                                           v["offsets"], sc, 64)
        assert [names[d - 1] for d in sorted(int(x) for x in hits["doc"])] == v["docs"]


def test_term_meta_codec_writer_and_reader_twins():
    """postings_writer::encode / postings_reader::decode of the term dictionary's
    stats records (formats_10.cpp:576-604, 3421-3456).  Two independent restatements of the
    WRITER (the emitter iresearch_amd/index/synth_dict.cpp and the oracle's orc_encode_term_meta)
    must produce the same bytes, and the oracle's restatement of the READER must give the metas
    back — for fields with and without positions / frequencies and the framing cases: a single
    doc (e_single_doc), exactly 128 docs (no skip pointer), 129 (skip pointer), pos_end present
    only when the term has more than 128 positions."""
    from iresearch_amd import synth
    rng = np.random.default_rng(9)
    n_docs = 5000

    def lst(n, tf_hi):
        d = np.sort(rng.choice(np.arange(1, n_docs + 1), n, replace=False)).astype(np.uint32)
        return d, rng.integers(1, tf_hi + 1, n).astype(np.uint32)

    # (docs, max tf): single doc with tf 1 / tf 200 (pos_end only for the second), 127/128/129
    # docs with tf == 1 (128 positions: no pos_end; 129: pos_end), long lists
    shapes = [(1, 1), (1, 200), (5, 1), (127, 1), (128, 1), (129, 1), (128, 3), (300, 2), (2000, 4)]
    plain = [lst(n, hi) for n, hi in shapes]
    with_pos = []
    for d, f in plain:
        pos = np.concatenate([np.sort(rng.choice(np.arange(1, 400), int(x), replace=False)) for x in f])
        with_pos.append((d, f, pos.astype(np.uint32)))
    for lists, has_pos in ((plain, False), (with_pos, True)):
        seg = synth.segment_from_lists(lists, n_docs, synth.LAYOUT_SIMD4)
        metas = seg.metas
        assert (metas["docs_count"] == [n for n, _ in shapes]).all()
        if has_pos:   # pos_end: set exactly where the reader will look for it
            for m in metas:
                assert (int(m["pos_end"]) != 0xFFFFFFFFFFFFFFFF) == (int(m["freq"]) > 128)
        stream = synth.term_meta_stream(metas, has_freq=True, has_pos=has_pos)
        assert np.array_equal(stream, oracle.encode_term_metas(metas, has_pos=has_pos))
        back = oracle.decode_term_metas(stream, len(metas), has_freq=True, has_pos=has_pos)
        for got, want in zip(back, metas):
            assert got["docs_count"] == want["docs_count"] and got["freq"] == want["freq"]
            assert got["doc_start"] == want["doc_start"]
            if has_pos:
                assert got["pos_start"] == want["pos_start"]
                assert got["pos_end"] == want["pos_end"]
            if want["docs_count"] == 1:
                assert int(got["e_skip_start"]) & 0xFFFFFFFF == int(want["e_skip_start"]) & 0xFFFFFFFF
            elif want["docs_count"] > 128:
                assert got["e_skip_start"] == want["e_skip_start"]
    # a field without frequencies: freq stays 0, no "freq - docs_count" record
    nofreq = np.zeros(3, synth.TERM_META)
    nofreq["docs_count"] = [1, 128, 500]
    nofreq["doc_start"] = [40, 40, 400]
    nofreq["e_skip_start"] = [7, 0, 333]
    nofreq["pos_end"] = np.uint64(0xFFFFFFFFFFFFFFFF)
    stream = synth.term_meta_stream(nofreq, has_freq=False)
    assert np.array_equal(stream, oracle.encode_term_metas(nofreq))
    back = oracle.decode_term_metas(stream, 3, has_freq=False)
    assert (back["docs_count"] == nofreq["docs_count"]).all() and (back["freq"] == 0).all()
    assert (back["doc_start"] == nofreq["doc_start"]).all()
    assert int(back[0]["e_skip_start"]) == 7 and int(back[2]["e_skip_start"]) == 333


def test_term_dictionary_and_columnstore_files():
    """the emitter's `.tm` (blocks written the way field_writer does) read by the
    oracle's restatement of the reference iterator, and its columnstore2 pair read by the
    oracle's column reader — nested blocks, floor blocks, 1/2/4-byte values, a column written
    block by block (fresh segment) and in one piece (consolidated)."""
    from iresearch_amd import synth
    seg = synth.build_segment(30_000, 2048)
    keep = [i for i in range(len(seg.metas)) if seg.metas[i]["docs_count"]]
    terms = [synth.term_bytes_of(i) for i in keep]
    tm, root = synth.term_dictionary(terms, seg.metas[keep])
    got_terms, got_metas = oracle.walk_term_dictionary(tm, root)
    assert got_terms == terms
    for g, w in zip(got_metas, seg.metas[keep]):
        assert g["docs_count"] == w["docs_count"] and g["freq"] == w["freq"]
        assert g["doc_start"] == w["doc_start"]
    with pytest.raises(ValueError):
        oracle.walk_term_dictionary(tm[:len(tm) // 2], root)
    rng = np.random.default_rng(4)
    for width in (1, 2, 4):
        n = 70_000 if width == 1 else 66_000     # more than one 65536-doc block
        vals = rng.integers(0, 256, n * width).astype(np.uint8)
        hdr = synth.norm2_header(width, 1, 200)
        for dense in (False, True):
            csd, csi, cid = synth.columnstore(vals, width, min_doc=1, payload=hdr, dense_fixed=dense)
            vb, mn, payload, values = oracle.read_fixed_column(csi, csd, cid)
            assert (vb, mn, payload) == (width, 1, hdr) and np.array_equal(values, vals)
        with pytest.raises(ValueError):
            oracle.read_fixed_column(csi, csd, cid + 1)   # a mask column: not a fixed-length one


def test_document_mask_file_writer_and_reader_twins():
    """`.doc_mask`: the emitter (DocumentMaskWriter::write, formats_10.cpp:3245-3268) against the
    oracle's reader (DocumentMaskReader::read :3275-3312); the product's host reader is the third
    (tests/cpp/test_segment.cpp).  And what the mask means to a search: MaskDocIterator."""
    rng = np.random.default_rng(4)
    for n in (0, 1, 127, 128, 5000):
        docs = rng.integers(1, 1 << 31, n).astype(np.uint32)
        f = synth.document_mask(docs)
        assert np.array_equal(oracle.read_document_mask(f), docs)
        if n:
            bad = f.copy()
            bad[len(bad) // 2] ^= 1
            with pytest.raises(ValueError):
                oracle.read_document_mask(bad)
    # a masked search = the unmasked one without the deleted docs (MaskDocIterator::next)
    seg = synth.build_segment(30_000, 64)
    gone = np.unique(rng.integers(1, 30_001, 3000).astype(np.uint32))
    import parity
    metas = parity.metas_for(seg, [0, 5, 9])
    sc = oracle.Scorer(oracle.SCORER_BM25, 1.2, 0.75, 0)
    plain = parity.oracle_view(seg)
    seg.doc_mask = gone
    masked = parity.oracle_view(seg)
    dwt = [int(m["docs_count"]) for m in metas]
    s0, m0 = oracle.score_all(plain, metas, oracle.OP_OR, sc, seg.docs_with_field, dwt, seg.total_term_freq)
    s1, m1 = oracle.score_all(masked, metas, oracle.OP_OR, sc, seg.docs_with_field, dwt, seg.total_term_freq)
    keep = np.ones(len(m0), bool)
    keep[gone] = False
    assert np.array_equal(m1.astype(bool), m0.astype(bool) & keep) and m1.sum() < m0.sum()
    assert np.array_equal(s1[keep], s0[keep]) and not s1[gone].any()


def test_scored_states_against_the_standard_heap(tmp_path):
    """limited_sample_collector keeps its scored states in a std::push_heap / pop_heap heap of
    indices; which of two EQUAL keys (the same docs_count at the same visit offset in two
    segments) gets replaced is decided inside those functions.  The oracle's restatement
    (oracle.scored_states) and the product's (search.scored_states) move the heap's elements as
    libstdc++ does: both against tests/cpp/collector_heap.cpp, the same container algorithm on
    the real std:: functions."""
    import subprocess
    from iresearch_amd import search
    exe = tmp_path / "collector_heap"
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", str(exe),
                    str(Path(__file__).parent / "cpp" / "collector_heap.cpp")], check=True)
    rng = np.random.default_rng(12)
    for _ in range(400):
        visits = [rng.integers(0, 5, int(rng.integers(0, 14))).tolist() for _ in range(int(rng.integers(1, 5)))]
        limit = int(rng.integers(0, 10))
        text = "%d %d\n" % (limit, len(visits)) + "".join(
            "%d %s\n" % (len(v), " ".join(map(str, v))) for v in visits)
        out = subprocess.run([str(exe)], input=text, capture_output=True, text=True, check=True).stdout
        want = [tuple(int(x) for x in line.split()) for line in out.splitlines()]
        assert oracle.scored_states(visits, limit) == want, (visits, limit)
        assert search.scored_states(visits, limit) == want, (visits, limit)
