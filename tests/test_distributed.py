"""N > 1 path on CPU: two processes (gloo), one segment each, global statistics,
all-gather of per-segment top-k, device-side merge (run here by the emulator
build) — compared with the oracle's single heap over both segments
(utils/index-search.cpp:719-779)."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
K = 50
N_SEGS = 4
DOCS = 12_000
MAX_RANK = 128


def _filters(phrase=False):
    from iresearch_amd import synth
    from iresearch_amd.search import And, Or, by_phrase, by_term
    if phrase:
        return [by_phrase([0, 1]), by_phrase([2, 0, 1]), by_phrase([5, 9]),
                by_phrase([0, 3], [0, 2]), by_phrase([1, 1])]
    ranks = synth.make_queries(5, 8, 2, MAX_RANK)
    fl = [Or([by_term(int(r) - 1) for r in row]) for row in ranks]
    fl += [by_term(7), Or([by_term(3), by_term(99)]), And([by_term(1), by_term(20)])]
    return fl


def _worker(rank, world, port, sim_path, out_dir, phrase=False):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from iresearch_amd import _lib, distributed, search, synth
    from iresearch_amd.search import BM25
    L = _lib.bind(C.CDLL(sim_path))
    my = distributed.segments_of_rank(N_SEGS, rank, world)
    segs = {s: synth.build_segment(DOCS, MAX_RANK, first_doc=s * DOCS, with_positions=phrase)
            for s in my}
    local = {s: (segs[s].docs_with_field, segs[s].total_term_freq,
                 np.asarray(segs[s].metas["docs_count"])) for s in my}
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    stats = {}
    for g in gathered:
        stats.update(g)
    seg_stats = [search.SegmentStats(*stats[s]) for s in range(N_SEGS)]
    filters = _filters(phrase)
    prep = search.prepare(filters, BM25(), seg_stats)
    nq = len(filters)
    # the production flow (bench.py): ONE batch over the rank's segments, results written
    # straight into its slots of the exchange buffer, one collective, device-side merge
    readers = [search.SegmentReader.from_synth(segs[s], L=L) for s in my]
    ex = distributed.TopkExchange(L, 0, N_SEGS, rank, world, nq, K, "cpu")
    b = search.QueryBatch(readers, prep, K).run()
    hp, cp = ex.slot(0)
    b.results_to_device(hp, cp)
    oh, osg, oc = ex.run()
    # ... and the one-shot form over per-segment batches must agree with it
    lists = []
    for i, r in enumerate(readers):
        sb = r.batch(prep, K).run()
        h = torch.zeros((nq, K), dtype=torch.int64)
        c = torch.zeros((nq,), dtype=torch.int32)
        sb.results_to_device(h.data_ptr(), c.data_ptr())
        lists.append((my[i], h, c))
        sb.close()
    oh2, osg2, oc2 = distributed.gather_merge(L, 0, lists, N_SEGS, rank, world, nq, K, "cpu")
    assert torch.equal(oh, oh2) and torch.equal(osg, osg2) and torch.equal(oc, oc2)
    # ... and so must bench.py's pipelined form (the collective of step i is in flight while
    # step i+1 runs; its merge comes one step later), on both buffer sets
    px = distributed.PipelinedExchange(L, 0, N_SEGS, rank, world, nq, K, "cpu")
    for it in range(3):
        b.run()
        prev = px.finish()
        if it:
            assert all(torch.equal(x, y) for x, y in zip(prev, (oh, osg, oc))), it
        b.results_to_device(*px.slot(it & 1, 0))
        px.start(it & 1)
    last = px.finish()
    assert all(torch.equal(x, y) for x, y in zip(last, (oh, osg, oc)))
    assert px.finish() is None
    # ... and with the collective behind the C ABI (irs_hip_comm_* / irs_hip_topk_allgather:
    # RCCL on the GPU, a shared-memory stand-in in this emulator build) instead of
    # torch.distributed: same answer, one-shot and pipelined
    comm = distributed.Communicator(L, 0, rank, world)
    ex2 = distributed.TopkExchange(L, 0, N_SEGS, rank, world, nq, K, "cpu", comm=comm)
    b.results_to_device(*ex2.slot(0))
    assert all(torch.equal(x, y) for x, y in zip(ex2.run(), (oh, osg, oc)))
    px2 = distributed.PipelinedExchange(L, 0, N_SEGS, rank, world, nq, K, "cpu", comm=comm)
    for it in range(2):
        b.run()
        px2.finish()
        b.results_to_device(*px2.slot(it & 1, 0))
        px2.start(it & 1)
    assert all(torch.equal(x, y) for x, y in zip(px2.finish(), (oh, osg, oc)))
    comm.close()
    np.save(os.path.join(out_dir, "hits_%d.npy" % rank), oh.numpy())
    np.save(os.path.join(out_dir, "segs_%d.npy" % rank), osg.numpy())
    np.save(os.path.join(out_dir, "counts_%d.npy" % rank), oc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_matches_oracle(simlib, tmp_path):
    sim_path = simlib._name
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, sim_path, str(tmp_path)), nprocs=2, join=True)
    import oracle
    import parity
    from iresearch_amd import _lib, synth
    from iresearch_amd.search import BM25
    r0 = [np.load(tmp_path / ("%s_0.npy" % n)) for n in ("hits", "segs", "counts")]
    r1 = [np.load(tmp_path / ("%s_1.npy" % n)) for n in ("hits", "segs", "counts")]
    for a, b in zip(r0, r1):                      # every rank holds the same merged result
        assert np.array_equal(a, b)
    hits = r0[0].view(_lib.HIT).reshape(r0[0].shape)
    segs_of = r0[1]
    counts = r0[2]
    segs = [synth.build_segment(DOCS, MAX_RANK, first_doc=s * DOCS) for s in range(N_SEGS)]
    filters = _filters()
    ref = parity.oracle_topk(segs, filters, BM25(), K)
    views = [parity.oracle_view(s) for s in segs]
    for q, (ohits, total) in enumerate(ref):
        n = int(counts[q])
        assert n == len(ohits)
        got = hits[q, :n]
        want = np.sort(ohits["score"])[::-1]
        assert np.allclose(got["score"], want, rtol=parity.REL_TOL, atol=0), q
        # ordered (score desc, segment asc, doc asc) — wand_test.cpp:72-86
        key = list(zip(-got["score"].astype(np.float64), segs_of[q, :n], got["doc"]))
        assert key == sorted(key), q
        assert (segs_of[q, :n] < N_SEGS).all()


def test_two_ranks_gloo_phrases_match_oracle(simlib, tmp_path):
    """The same flow for a batch of by_phrase queries: statistics of every phrase term are
    index-global, each rank runs one phrase batch over its segments, one all-gather, merge."""
    port = 29500 + ((os.getpid() * 3 + 101) % 2000)
    mp.spawn(_worker, args=(2, port, simlib._name, str(tmp_path), True), nprocs=2, join=True)
    import oracle
    import parity
    from iresearch_amd import _lib, synth
    from iresearch_amd.search import BM25
    r0 = [np.load(tmp_path / ("%s_0.npy" % n)) for n in ("hits", "segs", "counts")]
    r1 = [np.load(tmp_path / ("%s_1.npy" % n)) for n in ("hits", "segs", "counts")]
    for a, b in zip(r0, r1):
        assert np.array_equal(a, b)
    hits = r0[0].view(_lib.HIT).reshape(r0[0].shape)
    segs = [synth.build_segment(DOCS, MAX_RANK, first_doc=s * DOCS, with_positions=True)
            for s in range(N_SEGS)]
    views = [parity.oracle_view(s) for s in segs]
    osc = parity.oracle_scorer(BM25())
    some = 0
    for q, ph in enumerate(_filters(True)):
        metas = np.stack([parity.metas_for(s, ph.terms) for s in segs])
        ohits, total = oracle.search_phrase(views, metas, ph.offsets, osc, K, ph.boost)
        n = int(r0[2][q])
        assert n == len(ohits), (q, n, len(ohits))
        some += n
        got = hits[q, :n]
        want = np.sort(ohits["score"])[::-1]
        assert np.allclose(got["score"], want, rtol=parity.REL_TOL, atol=0), q
        key = list(zip(-got["score"].astype(np.float64), r0[1][q, :n], got["doc"]))
        assert key == sorted(key), q
    assert some > 0


def _misled_segment(si, tile=12288, stride=16, n_tiles=64):
    """cases.case_shared_threshold_misled's segment: the tiles the pilot samples (unit 0 of a
    rank's one local segment: phase 0) hold far better docs than the rest."""
    import cases
    from iresearch_amd import synth
    n_docs = tile * n_tiles
    rng = np.random.default_rng(31 + si)
    sampled = [t for t in range(n_tiles) if t % stride == 0]
    hot = np.concatenate([1 + t * tile + np.sort(rng.choice(tile, 400, replace=False))
                          for t in sampled]).astype(np.uint32)
    hot_f = rng.integers(1, 64, hot.size).astype(np.uint32)
    cold = np.sort(rng.choice(n_docs, 3000, replace=False)).astype(np.uint32) + 1
    return synth.segment_from_lists([(hot, hot_f), (cold, np.ones(cold.size, np.uint32))], n_docs,
                                    synth.LAYOUT_SIMD4, None)


def _threshold_worker(rank, world, port, sim_path, out_dir):
    """One threshold per query across ranks (irs_hip_batch_set_comm): the pilot histograms are
    summed over the ranks inside the run; the merged top k must not change by a bit."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import parity
    from iresearch_amd import _lib, distributed, search, synth
    from iresearch_amd.search import BM25, TFIDF, And, Or, by_term
    L = _lib.bind(C.CDLL(sim_path))
    comm = distributed.Communicator(L, 0, rank, world)

    def everyones(local):
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        out = {}
        for g in gathered:
            out.update(g)
        return out

    # ---- A: 4 segments of different sizes, 2 per rank; a term missing from one of them ----------
    sizes = (70_000, 30_000, 140_000, 50_000)
    first = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    my = distributed.segments_of_rank(N_SEGS, rank, world)
    segs = {s: synth.build_segment(int(sizes[s]), 256, first_doc=int(first[s])) for s in my}
    if 1 in segs:
        segs[1].metas[255]["docs_count"] = 0
    stats = everyones({s: parity.segment_stats(segs[s]) for s in my})
    ranks = synth.make_queries(6, 8, 12, 256, synth.SEED + 9)
    filters = [Or([by_term(int(r) - 1) for r in row]) for row in ranks]
    filters += [by_term(255), Or([by_term(20), by_term(255, 2.0)]),
                Or([by_term(12), by_term(13), by_term(14)], min_match=2), And([by_term(12), by_term(13)])]
    nq = len(filters)
    readers = [search.SegmentReader.from_synth(segs[s], L=L) for s in my]
    for tag, scorer in (("bm25", BM25()), ("tfidf", TFIDF(True))):
        prep = search.prepare(filters, scorer, [stats[s] for s in range(N_SEGS)])
        for k in (10, 300):
            merged, listed = [], []
            for across in (False, True):
                # (joined streams, whatever the cost rule makes of this small corpus)
                b = search.QueryBatch(readers, prep, k).set_shared_threshold(True)
                b.set_path(_lib.PATH_JOINED)
                if across:
                    b.set_comm(comm)
                b.run()
                ex = distributed.TopkExchange(L, 0, N_SEGS, rank, world, nq, k, "cpu")
                b.results_to_device(*ex.slot(0))
                assert b.reruns() == 0
                merged.append([t.clone() for t in ex.run()])
                listed.append(int(b.results()[1][:, :6].sum()))
                b.close()
            # bit for bit the merged result of per-rank thresholds
            assert all(torch.equal(x, y) for x, y in zip(*merged)), (tag, k)
            # ... and with the merged k-th score pushed back as irs::score::Min (the harness's heap
            # root, index-search.cpp:737-777) next to the cross-rank threshold: the same top k.  The
            # caller's scores are binned in the scale the group ends up with — a term missing from a
            # segment makes the group's bound larger than that unit's own (bins worked
            # out before the re-scaling dropped docs at or above the caller's score)
            oh, _, oc = merged[1]
            hv = oh.numpy().view(_lib.HIT)["score"].reshape(nq, k)
            kth = np.array([hv[q, int(oc[q]) - 1] if int(oc[q]) else 0.0 for q in range(nq)], np.float32)
            b = search.QueryBatch(readers, prep, k).set_shared_threshold(True).set_path(_lib.PATH_JOINED)
            b.set_comm(comm).set_min_scores(kth).run()
            ex = distributed.TopkExchange(L, 0, N_SEGS, rank, world, nq, k, "cpu")
            b.results_to_device(*ex.slot(0))
            again = [t.clone() for t in ex.run()]
            b.close()
            assert all(torch.equal(x, y) for x, y in zip(merged[1], again)), (tag, k, "score::Min")
            both = torch.tensor(listed)
            dist.all_reduce(both)
            if tag == "bm25" and k == 300:   # the ranks share the work of finding k docs
                assert int(both[1]) < int(both[0]), both
            if tag == "tfidf":               # (no score bound common to all segments: no groups)
                assert int(both[1]) >= int(both[0])
            if rank == 0:
                np.save(os.path.join(out_dir, "A_%s_%d_hits.npy" % (tag, k)), merged[1][0].numpy())
                np.save(os.path.join(out_dir, "A_%s_%d_counts.npy" % (tag, k)), merged[1][2].numpy())
    for r in readers:
        r.close()

    # ---- B: the pilot sample misleads the shared threshold on BOTH ranks: the group sums (also
    # summed over the ranks) make every rank re-run, together, with the sound threshold ----------
    seg = _misled_segment(rank)
    stats = everyones({rank: parity.segment_stats(seg)})
    sr = search.SegmentReader.from_synth(seg, L=L)
    filters = [by_term(0), Or([by_term(0), by_term(1)]), by_term(1)]
    k = 400
    prep = search.prepare(filters, BM25(1.2, 0.0), [stats[r] for r in range(world)])
    b = search.QueryBatch([sr], prep, k).configure(0, 16, 0).set_comm(comm)
    b.run()
    ex = distributed.TopkExchange(L, 0, world, rank, world, len(filters), k, "cpu")
    b.results_to_device(*ex.slot(0))
    assert b.reruns() == 1 and b.path() == _lib.PATH_JOINED
    oh, osg, oc = ex.run()
    if rank == 0:
        np.save(os.path.join(out_dir, "B_hits.npy"), oh.numpy())
        np.save(os.path.join(out_dir, "B_counts.npy"), oc.numpy())
    b.close()
    # ---- C: batch i is re-run WHILE batch i + 1 is already submitted (a serving loop pipelined one
    # deep: run(cur), then deliver(prev)).  Both batches issue collectives on the same communicator;
    # the recovery of i takes its turn in the library's per-device queue behind the run of i + 1 on
    # every rank alike (issued from the caller's thread it raced the worker's run)
    prep2 = search.prepare(filters[::-1], BM25(1.2, 0.0), [stats[r] for r in range(world)])
    alone = search.QueryBatch([sr], prep2, 50).set_comm(comm)
    alone.run()
    ex2 = distributed.TopkExchange(L, 0, world, rank, world, len(filters), 50, "cpu")
    alone.results_to_device(*ex2.slot(0))
    want2 = [t.clone() for t in ex2.run()]
    alone.close()
    for use_worker in (1, 0):
        b1 = search.QueryBatch([sr], prep, k).configure(0, 16, 0).set_comm(comm).set_async(use_worker)
        b2 = search.QueryBatch([sr], prep2, 50).set_comm(comm).set_async(use_worker)
        b1.run()
        if use_worker:
            b2.run()          # submitted before anybody looked at b1's status
        ex1 = distributed.TopkExchange(L, 0, world, rank, world, len(filters), k, "cpu")
        b1.results_to_device(*ex1.slot(0))      # verifies b1 -> collective re-run
        assert b1.reruns() == 1
        if not use_worker:
            b2.run()
        ex2 = distributed.TopkExchange(L, 0, world, rank, world, len(filters), 50, "cpu")
        b2.results_to_device(*ex2.slot(0))
        assert b2.reruns() == 0
        got1, got2 = ex1.run(), ex2.run()
        assert all(torch.equal(x, y) for x, y in zip(got1, (oh, osg, oc))), use_worker
        assert all(torch.equal(x, y) for x, y in zip(got2, want2)), use_worker
        b1.close()
        b2.close()
    sr.close()
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_one_threshold(simlib, tmp_path):
    """the harness keeps ONE heap over all segments (index-search.cpp:719-779);
    sharded over two ranks that is one threshold per query for both ranks' units.  The merged
    top k equals — bit for bit — the single-rank batch over all four segments, and the oracle's."""
    port = 29500 + ((os.getpid() * 5 + 211) % 2000)
    mp.spawn(_threshold_worker, args=(2, port, simlib._name, str(tmp_path)), nprocs=2, join=True)
    import cases  # noqa: F401  (sys.path side effects of tests/)
    import parity
    from iresearch_amd import _lib, search, synth
    from iresearch_amd.search import BM25, TFIDF, And, Or, by_term
    L = simlib
    sizes = (70_000, 30_000, 140_000, 50_000)
    first = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    segs = [synth.build_segment(int(n), 256, first_doc=int(f)) for n, f in zip(sizes, first)]
    segs[1].metas[255]["docs_count"] = 0
    readers = [search.SegmentReader.from_synth(s, L=L) for s in segs]
    ranks = synth.make_queries(6, 8, 12, 256, synth.SEED + 9)
    filters = [Or([by_term(int(r) - 1) for r in row]) for row in ranks]
    filters += [by_term(255), Or([by_term(20), by_term(255, 2.0)]),
                Or([by_term(12), by_term(13), by_term(14)], min_match=2), And([by_term(12), by_term(13)])]
    for tag, scorer in (("bm25", BM25()), ("tfidf", TFIDF(True))):
        prep = search.prepare(filters, scorer, [parity.segment_stats(s) for s in segs])
        for k in (10, 300):
            b = search.QueryBatch(readers, prep, k).set_shared_threshold(True)
            h, c, _ = b.set_path(_lib.PATH_JOINED).run().results()
            one = search.merge_topk_host([(h[i], c[i]) for i in range(len(segs))], k)
            got_h = np.load(tmp_path / ("A_%s_%d_hits.npy" % (tag, k))).view(_lib.HIT)
            got_c = np.load(tmp_path / ("A_%s_%d_counts.npy" % (tag, k)))
            for q, rows in enumerate(one):
                assert int(got_c[q]) == len(rows), (tag, k, q)
                two = got_h.reshape(len(filters), k)[q, :len(rows)]
                assert [r[0] for r in rows] == [float(x) for x in two["score"]], (tag, k, q)
                assert [r[2] for r in rows] == [int(x) for x in two["doc"]], (tag, k, q)
            b.close()
    for r in readers:
        r.close()
    # B against the oracle's heap over both segments
    segs = [_misled_segment(si) for si in range(2)]
    filters = [by_term(0), Or([by_term(0), by_term(1)]), by_term(1)]
    ref = parity.oracle_topk(segs, filters, BM25(1.2, 0.0), 400)
    got_h = np.load(tmp_path / "B_hits.npy").view(_lib.HIT).reshape(len(filters), 400)
    got_c = np.load(tmp_path / "B_counts.npy")
    for q, (ohits, total) in enumerate(ref):
        assert int(got_c[q]) == len(ohits), q
        assert np.allclose(got_h[q, :len(ohits)]["score"], np.sort(ohits["score"])[::-1],
                           rtol=parity.REL_TOL, atol=0), q


def test_bench_is_launchable_on_two_ranks(simlib):
    """bench.py under the driver's own multi-GPU launch line (torch.distributed.run, one
    process per rank, MASTER_ADDR 127.0.0.1), in its emulator dry-run mode: the whole
    multi-rank control flow — segment partition, global statistics, per-segment batches,
    one all-gather per step, merge, MAX-over-ranks timing, ONE JSON line from rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29500 + ((os.getpid() * 7 + 13) % 2000)
    env = dict(os.environ, IRS_BENCH_SIM=simlib._name, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--docs", "48000", "--queries", "5", "--k", "20", "--segments", "4", "--no-cpu"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["value"] > 0 and d["unit"] == "queries/s" and d["scaling"] == "strong"
    assert d["config"]["segments"] == 4 and d["config"]["reruns_rank0"] == 0
    assert d["roofline"]["launches_per_step"] == 1       # ONE batch over rank 0's 2 segments
    assert d["cpu_baseline"] is None
    # the exchange went through the C ABI's communicator and its self-test saw both ranks
    assert d["config"]["collective"].startswith("irs_hip_topk_allgather")
    assert d["config"]["ranks_seen"] == 2 and d["config"]["rccl_library"]
    # ... and the ranks shared one threshold per query (irs_hip_batch_set_comm)
    assert d["config"]["threshold"].startswith("one per query over all ranks")


def test_bench_config5_is_launchable_on_two_ranks(simlib):
    """`bench.py --config 5` (AND + by_phrase, TF-IDF, block-max WAND, segments with positions
    sharded over ranks) under the driver's launch line, emulator dry run: two batches per rank
    and step, two exchanges, ONE JSON line with the bytes actually decoded."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29500 + ((os.getpid() * 11 + 17) % 2000)
    env = dict(os.environ, IRS_BENCH_SIM=simlib._name, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--config", "5", "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--docs", "40000", "--queries", "6", "--k", "20", "--segments", "4"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    assert d["config"]["segments"] == 4 and d["config"]["queries_per_step"] == 12
    t = d["config"]["bytes_touched_per_step"]
    a = d["config"]["algorithmic_bytes_per_step"]
    assert 0 < t["and_doc_and_norm"] <= a["and"] and t["phrase_doc"] > 0
    assert d["config"]["collective"].startswith("irs_hip_topk_allgather")
    assert d["config"]["ranks_seen"] == 2 and d["config"]["rccl_library"]
