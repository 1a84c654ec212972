"""GPU tier: parity of the HIP path with the oracle through the C ABI on a real
MI355X.  Integer/byte/index work is compared bit-exactly; scores within 1e-5
relative (BASELINE.json north_star)."""
import numpy as np
import pytest

import cases
import parity
from iresearch_amd import search, synth
from iresearch_amd.search import BM25, Or, by_term

pytestmark = pytest.mark.gpu
LAYOUTS = [synth.LAYOUT_SIMD4, synth.LAYOUT_SCALAR]


@pytest.mark.parametrize("layout", LAYOUTS)
def test_decode_reference_lists(gpulib, layout):
    cases.case_decode_reference_lists(gpulib, layout)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_decode_sizes(gpulib, layout):
    cases.case_decode_sizes(gpulib, layout, sizes=(1, 2, 117, 127, 128, 129, 255, 256, 319, 1024,
                                                   10_000, 32_768))


@pytest.mark.parametrize("layout", LAYOUTS)
def test_decode_edge_blocks(gpulib, layout):
    cases.case_decode_edge_blocks(gpulib, layout)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_decode_synth(gpulib, layout):
    cases.case_decode_synth(gpulib, layout, 200_000, 1024)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_queries_all_scorers(gpulib, layout):
    cases.case_queries_all_scorers(gpulib, 300_000, 1024, layout)


def test_queries_tiles_and_strides(gpulib):
    cases.case_queries_tiles_and_strides(gpulib, 200_000, 512)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_queries_ragged(gpulib, layout):
    cases.case_queries_ragged(gpulib, layout)


def test_no_norms(gpulib):
    cases.case_no_norms(gpulib)


@pytest.mark.parametrize("width", [2, 4])
def test_wide_norms(gpulib, width):
    cases.case_wide_norms(gpulib, width)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_decode_without_freq(gpulib, layout):
    cases.case_decode_without_freq(gpulib, layout)


@pytest.mark.parametrize("layout", [0, 1])
def test_wand_data(gpulib, layout):
    cases.case_wand_data(gpulib, layout)


@pytest.mark.parametrize("layout", [0, 1])
def test_bit_union(gpulib, layout):
    cases.case_bit_union(gpulib, layout)
    cases.case_bit_union(gpulib, layout, has_freq=False)


def test_pilot_misled(gpulib):
    cases.case_pilot_misled(gpulib)


def test_multi_segment(gpulib):
    cases.case_multi_segment(gpulib, 600_000, 1024, n_segs=4, k=1000)


def test_merge_ties(gpulib):
    cases.case_merge_ties(gpulib)
    cases.case_merge_ties(gpulib, n_lists=8, nq=5, k=1000, seed=5)


def test_errors(gpulib):
    cases.case_errors(gpulib)


def test_config1_by_term_top10_100k(gpulib):
    """BASELINE config 1: single by_term BM25 top-10 on a 100k-doc index."""
    seg = synth.build_segment(100_000, 256)
    cases.run_and_check(gpulib, seg, [by_term(63)], BM25(), 10)


def test_config2_or2_top100_1m(gpulib):
    """BASELINE config 2: OR-of-2 BM25 top-100, 1M docs, 1 segment."""
    seg = synth.build_segment(1_000_000, 4096)
    ranks = synth.make_queries(40, 2, 16, 4096, synth.SEED + 1)
    filters = [Or([by_term(int(r) - 1) for r in row]) for row in ranks]
    cases.run_and_check(gpulib, seg, filters, BM25(), 100)


def test_config3_or8_top1000_subset(gpulib):
    """BASELINE config 3 shape (OR-of-8, top-1000) at 2M docs against the oracle,
    plus run-to-run determinism."""
    seg = synth.build_segment(2_000_000, 4096)
    ranks = synth.make_queries(24, 8, 16, 4096)
    filters = [Or([by_term(int(r) - 1) for r in row]) for row in ranks]
    sr = search.SegmentReader.from_synth(seg, L=gpulib)
    h1, c1, t1 = cases.run_and_check(gpulib, seg, filters, BM25(), 1000, sr=sr)
    h2, c2, t2 = cases.run_and_check(gpulib, seg, filters, BM25(), 1000, 8192, 4, sr=sr)
    assert np.array_equal(h1, h2) and np.array_equal(c1, c2) and np.array_equal(t1, t2)
    sr.close()
