"""CPU tier: the product sources (host code + kernels, unmodified) compiled against
the fiber emulator in tests/sim, driven through the C ABI, checked against the
oracle.  Small sizes; the same bodies run on the real GPU in test_gpu_parity.py."""
import pytest

import cases
from iresearch_amd import synth

LAYOUTS = [synth.LAYOUT_SIMD4, synth.LAYOUT_SCALAR]


@pytest.mark.parametrize("layout", LAYOUTS)
def test_decode_reference_lists(simlib, layout):
    cases.case_decode_reference_lists(simlib, layout)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_decode_reference_packed(simlib, layout):
    cases.case_decode_reference_packed(simlib, layout)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_decode_sizes(simlib, layout):
    cases.case_decode_sizes(simlib, layout)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_decode_edge_blocks(simlib, layout):
    cases.case_decode_edge_blocks(simlib, layout)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_decode_synth(simlib, layout):
    cases.case_decode_synth(simlib, layout, 20_000, 300, step=3)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_queries_all_scorers(simlib, layout):
    cases.case_queries_all_scorers(simlib, 30_000, 256, layout)


def test_queries_k_extremes(simlib):
    from iresearch_amd import _lib
    seg = synth.build_segment(20_000, 128, with_positions=True)
    for k in (1, _lib.MAX_K):
        cases.run_and_check(simlib, seg, cases.standard_filters(128), cases.BM25(), k)
        cases.run_phrases(simlib, seg, [cases.by_phrase([0, 1])], cases.BM25(), k)


def test_queries_tiles_and_strides(simlib):
    cases.case_queries_tiles_and_strides(simlib, 40_000, 256)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_queries_ragged(simlib, layout):
    cases.case_queries_ragged(simlib, layout)


def test_no_norms(simlib):
    cases.case_no_norms(simlib)


@pytest.mark.parametrize("width", [2, 4])
def test_wide_norms(simlib, width):
    cases.case_wide_norms(simlib, width)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_decode_without_freq(simlib, layout):
    cases.case_decode_without_freq(simlib, layout)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_header_chain(simlib, layout):
    cases.case_header_chain(simlib, layout)


@pytest.mark.parametrize("layout", [0, 1])
def test_wand_data(simlib, layout):
    cases.case_wand_data(simlib, layout)


@pytest.mark.parametrize("layout", [0, 1])
def test_bit_union(simlib, layout):
    cases.case_bit_union(simlib, layout)
    cases.case_bit_union(simlib, layout, has_freq=False)


def test_reference_score_orders(simlib):
    cases.case_reference_score_orders(simlib)


def test_boolean_reference_vectors(simlib):
    cases.case_boolean_reference_vectors(simlib)


def test_many_items(simlib):
    cases.case_many_items(simlib, 40_000)


@pytest.mark.parametrize("joined", [False, True])
def test_pilot_misled(simlib, joined):
    cases.case_pilot_misled(simlib, joined=joined)


def test_accumulator_switch(simlib):
    cases.case_accumulator_switch(simlib)


@pytest.mark.parametrize("layout", [0, 1])
def test_join_edge_blocks(simlib, layout):
    cases.case_join_edge_blocks(simlib, layout)


@pytest.mark.parametrize("layout", [0, 1])
def test_paths_agree(simlib, layout):
    cases.case_paths_agree(simlib, layout=layout)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_paired_tiles(simlib, layout):
    cases.case_paired_tiles(simlib, layout=layout)


def test_scored_multiterm_expansion(simlib):
    cases.case_scored_expansion(simlib, sizes=(30_000, 12_000), max_rank=384)


@pytest.mark.parametrize("layout", [0, 1])
def test_conjunctions_with_a_sparse_lead(simlib, layout):
    cases.case_conj_sparse_lead(simlib, layout=layout, n_docs=150_000)


@pytest.mark.parametrize("layout", [0, 1])
def test_deleted_documents(simlib, layout):
    cases.case_doc_mask(simlib, layout=layout, num_docs=40_000, max_rank=128)


def test_join_counts(simlib):
    cases.case_join_counts(simlib)


def test_join_counts_boundary(simlib):
    cases.case_join_counts_boundary(simlib)


def test_shared_threshold(simlib):
    cases.case_shared_threshold(simlib)
    cases.case_shared_threshold_misled(simlib)


def test_multi_segment(simlib):
    cases.case_multi_segment(simlib, 45_000, 256)


def test_multi_segment_batch(simlib):
    cases.case_multi_segment_batch(simlib)


def test_merge_ties(simlib):
    cases.case_merge_ties(simlib)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_decode_positions(simlib, layout):
    cases.case_decode_positions(simlib, layout)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_phrase_queries(simlib, layout):
    cases.case_phrase_queries(simlib, layout, 12_000)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_phrase_reference_vectors(simlib, layout):
    cases.case_phrase_reference_vectors(simlib, layout)


def test_phrase_ragged(simlib):
    cases.case_phrase_ragged(simlib)
    cases.case_phrase_ragged(simlib, synth.LAYOUT_SCALAR, one_based=True)


def test_phrase_fuzz(simlib):
    cases.case_phrase_fuzz(simlib, iters=12, seed=1)


def test_phrase_multi_segment(simlib):
    cases.case_phrase_multi_segment(simlib)


def test_phrase_errors(simlib):
    cases.case_phrase_errors(simlib)


def test_full_vocabulary(simlib):
    cases.case_full_vocabulary(simlib)


def test_legacy_norms(simlib):
    cases.case_legacy_norms(simlib)


def test_zero_boost(simlib):
    cases.case_zero_boost(simlib)


def test_max_and_min_score_merging(simlib):
    cases.case_merge_types(simlib)


def test_min_score_pushdown(simlib):
    cases.case_min_score_pushdown(simlib)


def test_plan_ahead(simlib):
    cases.case_plan_ahead(simlib)


def test_host_results(simlib):
    cases.case_host_results(simlib)


def test_fresh_batches_and_trim(simlib):
    cases.case_fresh_batches_and_trim(simlib)


def test_wand_equals_exhaustive(simlib):
    cases.case_wand_equals_exhaustive(simlib)


def test_errors(simlib):
    cases.case_errors(simlib)
