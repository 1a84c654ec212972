#!/usr/bin/env python3
"""bench.py — queries/sec of the MI355X query path on BASELINE.json's headline
workload: OR-of-8-terms BM25 top-1000 on a synthetic 10M-doc Zipfian index.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A *step* is one pass of the hot path over one batch of 1000 queries (plan ->
pilot -> score -> select, plus the RCCL top-k all-gather + merge when N > 1).
N = 1 is BASELINE config 3 (one 10M-doc segment on one GPU); N > 1 is config 4:
the same 10M docs split into 8 segments, 8/N per GPU, global statistics, one
all-gather of per-segment top-k per step ("strong" scaling: total work fixed).

Rank 0 prints ONE JSON line.  `value` is whole-job queries/sec with the index
already resident in HBM.  `roofline` prices the dominant kernel (k_score)
against HBM bandwidth using ALGORITHMIC bytes and its average
duration measured with HIP events on the launch stream.  `cpu_baseline` is the
oracle's restatement of utils/index-search timed on this host (N = 1, rank 0).
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# ... and what a read-only stream of 16 B per lane really draws on this part
# (tools/micro/fetch_calib.hip, profiles/r04_fetch_calib.txt: 6.3 - 6.4 TB/s; 4 B per lane: 4.06)
HBM_MEASURED_GBS = 6300.0


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def usable_cpus():
    """CPUs this process can actually use: the affinity mask, capped by the container's CFS
    quota (cgroup v2 cpu.max / v1 cpu.cfs_quota_us).  Running more threads than that only
    gets them throttled."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(seg, ranks, k, seconds_hint=10.0):
    """Oracle (port of the index-search loop) on a bounded sample of the same queries, built
    -O3 -march=native on this host, timed with ONE thread and with as many
    threads as this container may use.  Imports oracle/ here and only here."""
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity
    native = oracle.use_native()
    cores = usable_cpus()
    view = parity.oracle_view(seg)
    sc = oracle.Scorer(oracle.SCORER_BM25, 1.2, 0.75, 0)

    def metas_of(rows):
        return np.stack([parity.metas_for(seg, [int(r) - 1 for r in row])[None] for row in rows])

    def timed(threads, hint):
        # calibrate on a few queries per thread, then size the sample (the bench queries,
        # cycled if there are too few) for ~hint seconds of wall time
        probe = ranks[: min(len(ranks), max(4, 4 * threads))]
        t0 = time.perf_counter()
        oracle.search_batch([view], metas_of(probe), oracle.OP_OR, sc, k, threads)
        dt = max(time.perf_counter() - t0, 1e-6)
        n = int(min(20000, max(len(probe), len(probe) * hint / dt)))
        sample = ranks[np.arange(n) % len(ranks)]
        t0 = time.perf_counter()
        oracle.search_batch([view], metas_of(sample), oracle.OP_OR, sc, k, threads)
        dt = time.perf_counter() - t0
        return n / dt, n, dt

    q1, n1, dt1 = timed(1, 0.6 * seconds_hint)
    qt, nt, dtt = timed(cores, seconds_hint)
    hw = os.cpu_count() or cores
    return {"value": round(qt, 3), "value_1thread": round(q1, 3), "unit": "queries/s",
            "cores": cores, "kind": "port",
            "flags": "-O3 -march=native" if native else "-O2 (native build failed)",
            "sample": "%d queries on %d threads in %.1f s, %d queries on 1 thread in %.1f s (the %d "
                      "bench queries, cycled); threads pop one task queue (index-search "
                      "--threads); %d = the CPUs this container may use (CFS quota) of %d "
                      "hardware threads" % (nt, cores, dtt, n1, dt1, len(ranks), cores, hw)}


def cpu_baseline_config5(segs, ands, phrases, scorer, k, seconds_hint=12.0):
    """Config 5 on the host: the oracle's restatement of the index-search loop over ALL segments
    (one heap per query, as the harness keeps it) on a bounded, interleaved sample of the same
    AND and by_phrase queries; the container's CPUs pop one task queue (index-search --threads).
    Imports oracle/ here and only here."""
    import concurrent.futures as cf

    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity
    native = oracle.use_native()
    cores = usable_cpus()
    views = [parity.oracle_view(sg) for sg in segs]
    osc = parity.oracle_scorer(scorer)

    def run_and(flt):
        terms = [t.term for t in flt.subs]
        metas = np.stack([parity.metas_for(sg, terms) for sg in segs])
        oracle.search(views, metas, oracle.OP_AND, osc, k, [t.boost for t in flt.subs])

    def run_phrase(ph):
        metas = np.stack([parity.metas_for(sg, ph.terms) for sg in segs])
        oracle.search_phrase(views, metas, ph.offsets, osc, k, ph.boost)

    tasks = []
    for a, ph in zip(ands, phrases):
        tasks += [(run_and, a), (run_phrase, ph)]

    def timed(threads, n):
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda t: t[0](t[1]), tasks[:n]))
        return time.perf_counter() - t0

    probe = min(len(tasks), 2 * cores)
    dt = max(timed(cores, probe), 1e-6)
    n = int(min(len(tasks), max(probe, probe * seconds_hint / dt)))
    dt = timed(cores, n)
    return {"value": round(n / dt, 3), "unit": "queries/s", "cores": cores, "kind": "port",
            "flags": "-O3 -march=native" if native else "-O2 (native build failed)",
            "sample": "%d queries (AND and by_phrase alternating, the bench's own) over %d segments "
                      "on %d threads in %.1f s" % (n, len(segs), cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--terms", type=int, default=8)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--segments", type=int, default=8, help="segments of the index when --gpus > 1")
    ap.add_argument("--max-rank", type=int, default=4096,
                    help="term ranks that get posting lists (the queries draw from [16, 4096] "
                         "whatever this is; 1048576 = the corpus's whole vocabulary: term tables, "
                         "block directory and open time at a real dictionary's size)")
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--stride", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--two-streams", action="store_true",
                    help="config 5: queue the AND batch and the phrase batch on a stream each (measured: "
                         "41.3 against 42.0 ms per step, DESIGN.md §3.12; the stage times then overlap)")
    ap.add_argument("--config", type=int, default=3, choices=[3, 5],
                    help="3 (default, with --gpus > 1: config 4): OR-of-8 BM25 top-1000 on 10 M docs; "
                         "5: AND-of-2..4 + 2-word by_phrase, TF-IDF, block-max WAND, on --docs "
                         "(default 50 M) docs in 8 segments with positions, 8 / N per GPU")
    ap.add_argument("--query-sets", type=int, default=0,
                    help="0 (default): every step executes a batch of queries that never ran before "
                         "(created, run and destroyed inside the step); N > 0: the old protocol — N "
                         "persistent batches of the same distribution replayed in rotation (A/B)")
    ap.add_argument("--no-cross-rank-threshold", action="store_true",
                    help="N > 1: every rank keeps the threshold of its own segments (A/B against "
                         "irs_hip_batch_set_comm: one threshold per query over all ranks)")
    ap.add_argument("--no-shared-threshold", action="store_true",
                    help="A/B: every (segment, query) unit of a rank's batch keeps its own threshold")
    ap.add_argument("--per-segment-batches", action="store_true",
                    help="one batch per local segment instead of one batch over all of them (A/B)")
    ap.add_argument("--force-segments", action="store_true",
                    help="use the multi-segment path (merge kernel) even on one GPU")
    ap.add_argument("--tasks", action="store_true",
                    help="replay the reference harness's task classes (scripts/iresearch-benchmark.tasks: "
                         "HighTerm ... MinMatch2High2Med) through the C ABI and through the restated CPU "
                         "loop, one line per category; not the headline metric")
    ap.add_argument("--tasks-file", default=None,
                    help="a task list in the harness's grammar (default: the reference's own, "
                         "tests/golden/benchmark_tasks.json)")
    ap.add_argument("--scorer", default="bm25", choices=["bm25", "tfidf"])
    args = ap.parse_args()
    if args.tasks:
        return main_tasks(args)
    if args.config == 5:
        return main_config5(args)

    import torch
    import torch.distributed as dist

    from iresearch_amd import _lib, distributed, search, synth
    from iresearch_amd.search import BM25, Or, by_term

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # IRS_BENCH_SIM=<path of tests/sim/libirs_hip_sim.so>: CONTROL-FLOW dry run for the CPU
    # test tier (tests/test_distributed.py) — the emulator library, CPU tensors, gloo instead
    # of RCCL.  Its timings mean nothing; it exists so that the multi-rank path the driver
    # launches is executed before it ever reaches an 8-GPU node.
    sim = os.environ.get("IRS_BENCH_SIM")
    if sim:
        import ctypes
        dev = torch.device("cpu")
        L = _lib.bind(ctypes.CDLL(sim))
        local_rank = 0
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        L = _lib.lib()
    sync = (lambda: None) if sim else torch.cuda.synchronize
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if sim:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    if not sim:
        torch.zeros(1, device=dev)   # (the context exists before the first timed open)
        sync()

    # ---- index: built on the host, staged to HBM once (not timed) ------------
    t0 = time.perf_counter()
    multi = world > 1 or args.force_segments
    if not multi:
        n_segments = 1
        my = [0]
    else:
        n_segments = args.segments
        my = distributed.segments_of_rank(n_segments, rank, world)
    per = args.docs // n_segments
    segs = {}
    for s in my:
        n = per if s < n_segments - 1 else args.docs - per * (n_segments - 1)
        segs[s] = synth.build_segment(n, args.max_rank, first_doc=s * per)
    log("built %d segment(s) in %.1f s" % (len(my), time.perf_counter() - t0))
    local_stats = {s: (segs[s].docs_with_field, segs[s].total_term_freq,
                       np.asarray(segs[s].metas["docs_count"])) for s in my}
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, local_stats)
        all_stats = {}
        for g in gathered:
            all_stats.update(g)
    else:
        all_stats = local_stats
    seg_stats = [search.SegmentStats(*all_stats[s]) for s in range(n_segments)]

    t0 = time.perf_counter()
    readers = {s: search.SegmentReader.from_synth(segs[s], device=local_rank, L=L) for s in my}
    sync()
    open_s = time.perf_counter() - t0
    open_bytes = sum(segs[s].doc_file.size + segs[s].norms.size for s in my)
    log("staged to HBM in %.2f s (%.1f MB resident on rank 0)" %
        (open_s, sum(r.device_bytes() for r in readers.values()) / 1e6))

    # ---- queries ---------------------------------------------------------------
    # A step executes a batch of queries THAT HAS NEVER RUN: new term rows (their own seed; set 0
    # = BASELINE config 3, seed + 2), prepared (index-global statistics -> scorer constants,
    # by_term::prepare + BM25::collect), turned into a batch (irs_hip_batch_create_multi: the
    # per-query records, the batch's distinct posting streams, the work lists of the kernels),
    # run, checked, delivered, destroyed — all inside the step, as the reference's harness builds
    # and prepares the filter and constructs the iterators inside its timers
    # (index-search.cpp:694-722).  Only the random term rows themselves are drawn ahead.
    # --query-sets N > 0: the round-1..3 protocol (N persistent batches replayed in rotation).
    replay = args.query_sets > 0
    n_probe = 3            # isolated (unpipelined) fresh batches, timed one by one
    n_host = 0 if world > 1 else max(2, min(args.steps, 20))
    n_rows = max(1, args.query_sets) if replay else min(64, n_probe + args.warmup + args.steps + n_host + 2)
    rank_sets = [synth.make_queries(args.queries, args.terms, 16, 4096,
                                    synth.SEED + 2 + (10 + i if i else 0)) for i in range(n_rows)]
    ranks = rank_sets[0]
    scorer = BM25()
    # ONE batch per rank over all of its segments (irs_hip_batch_create_multi): every kernel is
    # launched once for all (segment, query) pairs.  --per-segment-batches: one batch per
    # segment, back to back (A/B).
    if len(my) > 1 and not args.per_segment_batches:
        groups = {my[0]: list(my)}
    else:
        groups = {s: [s] for s in my}
    nq, k = args.queries, args.k

    def make_batches(i):
        rows = rank_sets[i % n_rows].astype(np.int64) - 1      # rank r is term ordinal r - 1
        out = {}
        for lead, members in groups.items():
            rs = [readers[s] for s in members]
            arrays = search.prepare_disjunctions(rows, scorer, seg_stats, rs, k)
            b = search.QueryBatch(rs if len(members) > 1 else rs[0], arrays)
            if args.tile or args.stride:
                b.configure(args.tile, args.stride, 0)
            if len(members) > 1 and not args.no_shared_threshold:
                # the per-segment lists are merged (all-gather + k_merge_topk): the segments of
                # a rank look for a query's k best docs TOGETHER (irs_hip_batch_set_shared_threshold)
                b.set_shared_threshold(True)
            if comm_thr is not None:
                # ... and the ranks TOGETHER: one threshold per query over all segments of the
                # index, as the harness's one heap has (irs_hip_batch_set_comm: the pilot
                # histograms are summed over the ranks inside the run)
                b.set_comm(comm_thr)
            b.profile(True)
            out[lead] = b
        return out

    # the collectives go through the library's own RCCL communicators (irs_hip_comm_*); torch
    # only carries their 128-byte ids to the other ranks.  Two of them: the top-k exchange of
    # step i overlaps the batch of step i+1, whose threshold all-reduces need a communicator
    # nothing else is using
    comm = comm_thr = None
    if world > 1:   # (on every rank or on none; torch.distributed remains the way out)
        comm = distributed.agreed_communicator(L, local_rank, rank, world, dev, log)
        per_rank = (n_segments + world - 1) // world
        every_rank_has_one = (world - 1) * per_rank < n_segments
        if (comm is not None and every_rank_has_one and len(groups) == 1
                and not args.no_cross_rank_threshold and not args.no_shared_threshold):
            comm_thr = distributed.agreed_communicator(L, local_rank, rank, world, dev, log)
    batch_sets = [make_batches(i) for i in range(n_rows)] if replay else None
    sptr = None if sim else C_void(torch.cuda.current_stream(dev).cuda_stream)
    # every buffer of the exchange step is allocated once (twice: two sets alternate); each
    # local segment's results are written straight into its slot of the send buffer
    exchange = distributed.PipelinedExchange(L, local_rank, n_segments if multi else 1, rank,
                                             world, nq, k, dev, comm=comm)
    # a multi-segment batch writes [segment][query][k] hits and [segment][query] counts: exactly
    # consecutive slots of the send buffer
    slots = [{lead: exchange.slot(ph, my.index(lead)) for lead in groups} for ph in (0, 1)]
    state = {"it": 0, "work": [], "joined": True, "paired": True, "reruns": 0, "retire": [], "host_hits": 0}

    def deliver(prev):
        # a finished step: checked (irs_hip_batch_results_to_device waits for THAT batch's own
        # event and reads its status word from page-locked memory; it re-runs the batch if a
        # threshold estimate or the candidate buffer fell short — irs_hip_batch_reruns counts
        # those), its per-segment top-k copied into the exchange slots, the all-gather started.
        cur, ph, host_results = prev
        if multi:
            exchange.finish(sptr)
        for s in cur:
            cur[s].results_to_device(slots[ph][s][0], slots[ph][s][1], sptr)
            if host_results:
                # ... and to page-locked host memory, where the reference's harness ends
                # (index-search.cpp:782-807): queued on the library's download stream behind THIS
                # batch's kernels only — the copy crosses PCIe while the next batch executes; the
                # arrays are read (irs_hip_batch_host_results) when the batch is retired
                cur[s].results_to_host()
        if rank == 0 and state.get("collect") is not None:
            # per-kernel HIP-event timings of that step (the batch's own events: no stream sync)
            state["collect"].append(np.sum([cur[s].timings() for s in cur], axis=0))
            state["work"].append(np.sum([cur[s].work() for s in cur], axis=0))
        state["joined"] = state["joined"] and all(b.path() == _lib.PATH_JOINED for b in cur.values())
        state["paired"] = state["paired"] and all(b.paired_tiles() for b in cur.values())
        if multi:
            exchange.start(ph)
        if not replay:
            # a delivered batch is destroyed one step LATER: its result copies were queued behind
            # the next step's kernels, and irs_hip_batch_destroy waits for them (its buffers go
            # back to the library's pool) — by then they are through
            retire()
            state["retire"] = [(b, host_results) for b in cur.values()]

    def retire():
        for b, host_results in state["retire"]:
            if host_results:
                hits, counts, totals = b.host_results()   # waits for that batch's copy
                state["host_hits"] += int(counts.sum())   # (the arrays are read)
            state["reruns"] += b.reruns()
            b.close()
        state["retire"] = []

    def step(host_results=False):
        # Steps are software-pipelined one deep: the batch of step i is prepared, created and
        # its kernels enqueued FIRST, then step i-1 is verified and delivered — the host builds
        # step i while the GPU still runs step i-1, and waits on step i-1's event while the GPU
        # already runs step i: no per-step host/device round trip on the critical path.  Every
        # step still ends (one step later; flush() for the last) with a checked,
        # device-resident top-k.  host_results: the hits also go to page-locked host memory
        # (irs_hip_batch_results_to_host), pipelined like everything else.
        ph = state["it"] & 1
        cur = batch_sets[state["it"] % n_rows] if replay else make_batches(state["it"])
        state["it"] += 1
        prev = state.get("prev")
        if prev is not None and prev[0] is cur:
            deliver(prev)     # the same batch again (replay of one set): its results go out first
            prev = None
        for s in cur:
            cur[s].run(sptr)
        if prev is not None:
            deliver(prev)
        state["prev"] = (cur, ph, host_results)

    def flush():
        prev = state.pop("prev", None)
        if prev is not None:
            deliver(prev)
        out = exchange.finish(sptr) if multi else None
        if not replay:
            sync()
            retire()
        return out

    def total_reruns():
        if replay:
            return sum(b[s].reruns() for b in batch_sets for s in b)
        return state["reruns"]

    # isolated executions of never-seen batches (no pipelining: host preparation + kernels +
    # verification in sequence), timed one by one — includes whatever re-run a misled threshold
    # estimate costs; the first also pays the allocations the pool keeps from then on
    first_ms = []
    for _ in range(n_rows if replay else n_probe):
        sync()
        t0 = time.perf_counter()
        step()
        flush()
        sync()
        first_ms.append(1e3 * (time.perf_counter() - t0))
    if replay:
        state["it"] = 0
    for _ in range(args.warmup):
        step()
    flush()
    sync()
    reruns_warm = total_reruns()
    score_ms = []
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    state["collect"] = score_ms
    state["work"] = []
    for _ in range(args.steps):
        step()
    flush()
    state["collect"] = None
    sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    reruns_timed = total_reruns() - reruns_warm
    # the same steps with the hits copied to host memory in every step (not `value`)
    host_elapsed = None
    if n_host:
        for _ in range(2):               # (warm-up: the page-locked result buffers enter the pool)
            step(host_results=True)
        flush()
        sync()
        t0 = time.perf_counter()
        for _ in range(n_host):
            step(host_results=True)
        flush()
        sync()
        host_elapsed = (time.perf_counter() - t0) / n_host
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    reruns = total_reruns()
    # per step, averaged over the timed steps (rank 0 collected them; the other ranks' shares
    # are summed below)
    if rank == 0 and state["work"]:
        alg_bytes, postings = [float(x) for x in np.mean(np.array(state["work"], dtype=np.float64), axis=0)]
    else:
        wb = batch_sets[0] if replay else make_batches(0)
        alg_bytes, postings = [float(x) for x in np.sum([wb[s].work() for s in wb], axis=0)]
        if not replay:
            for b in wb.values():
                b.close()
    batches = groups
    joined = state["joined"]
    paired = joined and state["paired"]
    rank0_alg_bytes = alg_bytes
    if world > 1:
        t = torch.tensor([alg_bytes, postings], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        alg_bytes, postings = float(t[0].item()), float(t[1].item())

    out = None
    if rank == 0:
        qps = args.steps * nq / elapsed
        roof = None
        if score_ms:
            ms = np.array(score_ms)                    # [steps][K_COUNT]
            avg = ms.mean(axis=0)
            tr = measured_traffic()
            if joined:
                # joined posting streams: the algorithmic bytes of a step flow through TWO
                # kernels — k_join decodes every distinct term of the batch once (bit-exact doc
                # ids + frequencies, norm join), k_join_score accumulates them per query.  The
                # roofline is priced on BOTH durations (the decode stage is part of the work);
                # k_join_score alone, the dominant kernel, is given next to it.
                stage_ms = float(avg[_lib.K_PLAN] + avg[_lib.K_SCORE])
                achieved = rank0_alg_bytes / (stage_ms * 1e-3) / 1e9
                names = _lib.KERNEL_NAMES_JOINED
                # (paired doc tiles: the score stage is k_join_score<2> — two tiles per visit in
                # 16-bit halves — followed by k_join_rescore, the exact sums of the docs it picked)
                kernel = "k_join + k_join_score + k_join_rescore" if paired else "k_join + k_join_score"
            else:
                # k_score on rank 0: its algorithmic bytes / its summed launch time per step
                stage_ms = float(avg[_lib.K_SCORE])
                achieved = rank0_alg_bytes / (stage_ms * 1e-3) / 1e9
                names = _lib.KERNEL_NAMES
                kernel = "k_score"
            roof = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "peak_measured": HBM_MEASURED_GBS,
                    "peak_measured_source": "tools/micro/fetch_calib.hip: read-only stream, 16 B per lane "
                                            "(profiles/r04_fetch_calib.txt)",
                    "frac_of_measured": round(achieved / HBM_MEASURED_GBS, 5),
                    # (no fraction of its own for the larger of the two kernels: the algorithmic
                    # bytes are consumed by the decode stage, not by it)
                    "dominant_kernel": {"name": ("k_join_score<2> + k_join_rescore (one stage)" if paired
                                                 else "k_join_score") if joined else "k_score",
                                        "avg_launch_ms": round(float(avg[_lib.K_SCORE]), 4)},
                    # HBM-side bytes per step of those kernels (PMC: 2 x FETCH_SIZE + WRITE_SIZE,
                    # separate rocprofv3 passes of this command at these kernel sources), else null
                    "traffic": None if (multi or tr is None) else int(tr["bytes"]),
                    "traffic_detail": None if multi else tr,
                    "algorithmic_bytes_per_launch": int(rank0_alg_bytes / len(batches)),
                    "launches_per_step": len(batches),
                    "avg_launch_ms": round(stage_ms, 4),
                    "kernel_ms": {n: round(float(v), 4) for n, v in zip(names, avg)}}
        out = {
            "metric": "queries/sec BM25 top-1000, OR-8-terms, 10M-doc Zipfian index @1/2/4/8 GPU",
            "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "value_with_results_on_host": None if host_elapsed is None else round(nq / host_elapsed, 2),
            "dtype": "u32+f32", "data": "synthetic" if not sim else "synthetic (EMULATOR DRY RUN)",
            "config": {
                "workload": "OR-of-%d terms BM25 top-%d, %d-doc Zipfian index, %d segment(s), "
                            "%d queries/step" % (args.terms, k, args.docs, n_segments, nq),
                "segments": n_segments, "queries_per_step": nq, "layout": "1_5simd",
                "indexed_ranks": args.max_rank,
                # irs_hip_segment_open of this rank's segments (file bytes over PCIe + the block
                # directory, packed image and tail tables built on the GPU), once per segment
                "segment_open": {"seconds": round(open_s, 3), "file_bytes": int(open_bytes),
                                 "GB_per_s": round(open_bytes / open_s / 1e9, 3)},
                "path": ("joined posting streams (k_join once per distinct term of the batch, "
                         "k_join_score per query%s)" % (" on paired doc tiles, k_join_rescore for the exact "
                                                        "sums of the docs it picks" if paired else "")) if joined
                        else "work items (every query decodes its own blocks)",
                "postings_per_step": int(postings), "algorithmic_bytes_per_step": int(alg_bytes),
                "batches": ("%d persistent batches replayed in rotation (--query-sets)" % n_rows) if replay
                           else "every step prepares, creates, runs and destroys a batch of queries "
                                "that never ran before (%d distinct query sets)" % n_rows,
                "query_sets": n_rows,
                # isolated (unpipelined) never-seen batches: host preparation + kernels + check
                "first_run_ms_per_set": [round(x, 3) for x in first_ms],
                "reruns_rank0": int(reruns), "reruns_in_timed_steps": int(reruns_timed),
                "parallelism": ("%d segments over %d GPU(s) + RCCL all-gather of per-segment "
                                "top-k + GPU merge" % (n_segments, world)) if multi
                               else "1 segment on 1 GPU",
                "collective": None if world == 1 else
                              ("irs_hip_topk_allgather (RCCL behind the C ABI)" if comm is not None
                               else "torch.distributed all_gather_into_tensor"),
                "rccl_library": None if comm is None else comm.library(),
                "ranks_seen": None if comm is None else comm.ranks_seen,
                "threshold": None if world == 1 else
                             ("one per query over all ranks (irs_hip_batch_set_comm: 2 all-reduces per batch)"
                              if comm_thr is not None else "one per query and rank")},
            "roofline": roof,
        }
    if rank == 0 and not multi and not args.no_cpu:
        for bs in (batch_sets or []):
            for b in bs.values():
                b.close()
        out["cpu_baseline"] = cpu_baseline(segs[0], ranks, k)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main_config5(args):
    """BASELINE config 5: AND-of-2..4 + 2-word by_phrase, TF-IDF (no norms => MaxFreq wand data,
    tfidf.cpp:364-386), block-max WAND, a 50 M-doc index with positions in 8 segments, 8 / N per
    GPU.  A step = one batch of --queries AND queries + one batch of --queries phrase queries
    over the rank's segments; the per-segment top-k lists are exchanged and merged as in config 4.
    Reports A(q) (exhaustive algorithmic bytes) AND the bytes the kernels really decoded."""
    import torch
    import torch.distributed as dist

    from iresearch_amd import _lib, distributed, search, synth
    from iresearch_amd.search import TFIDF, And, by_phrase, by_term
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    sim = os.environ.get("IRS_BENCH_SIM")   # control-flow dry run on the emulator (see main())
    if sim:
        import ctypes
        dev = torch.device("cpu")
        L = _lib.bind(ctypes.CDLL(sim))
        local_rank = 0
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        L = _lib.lib()
    sync = (lambda: None) if sim else torch.cuda.synchronize
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if sim:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    docs = args.docs if args.docs != 10_000_000 else 50_000_000
    n_segments = args.segments
    my = distributed.segments_of_rank(n_segments, rank, world)
    per = docs // n_segments
    t0 = time.perf_counter()
    # the field is indexed WITH its scorer, as a WAND-enabled IResearch index is: TF-IDF without
    # norms asks for MaxFreq wand data (tfidf.cpp:364-386) — the block-max pairs the pruning uses
    # are then read from the index's own skip entries (k_wand_skip0)
    segs = {s: synth.build_segment(per if s < n_segments - 1 else docs - per * (n_segments - 1),
                                   4096, first_doc=s * per, with_positions=True, wand_count=1,
                                   wand_kind=synth.WAND_MAX_FREQ) for s in my}
    log("built %d segment(s) with positions in %.1f s" % (len(my), time.perf_counter() - t0))
    local_stats = {s: (segs[s].docs_with_field, segs[s].total_term_freq,
                       np.asarray(segs[s].metas["docs_count"])) for s in my}
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, local_stats)
        all_stats = {}
        for g in gathered:
            all_stats.update(g)
    else:
        all_stats = local_stats
    seg_stats = [search.SegmentStats(*all_stats[s]) for s in range(n_segments)]
    readers = [search.SegmentReader.from_synth(segs[s], device=local_rank, L=L) for s in my]
    log("staged: %.1f MB resident on rank 0" % (sum(r.device_bytes() for r in readers) / 1e6))
    nq, k = args.queries, args.k if args.k != 1000 else 100
    sc = TFIDF(False)
    # Query sets (set 0 = the configuration's own seeds): a step creates its two batches anew
    # from the prepared arrays of the next set, runs them and destroys them — no batch object
    # is executed twice (the filters' statistics are collected ahead, once per set).
    n_sets = max(1, min(4, args.query_sets if args.query_sets > 0 else 4))
    sets = []
    for i in range(n_sets):
        ands_i = []
        for n_terms in (2, 3, 4):
            for row in synth.make_queries((nq + 2) // 3, n_terms, 16, 4096,
                                          synth.SEED + 5 + n_terms + 100 * i):
                ands_i.append(And([by_term(int(r) - 1) for r in row]))
        ands_i = ands_i[:nq]
        phrases_i = [by_phrase([int(r) - 1 for r in row])
                     for row in synth.make_queries(nq, 2, 16, 4096, synth.SEED + 9 + 100 * i)]
        arrays = {name: search.QueryArrays.from_prepared(readers, search.prepare(fl, sc, seg_stats), k)
                  for name, fl in (("and", ands_i), ("phrase", phrases_i))}
        sets.append((ands_i, phrases_i, arrays))
    ands, phrases = sets[0][0], sets[0][1]

    def make_batches(i):
        out = {}
        for name, arrays in sets[i % n_sets][2].items():
            b = search.QueryBatch(readers if len(readers) > 1 else readers[0], arrays)
            if name == "and":
                b.set_wand(True)
            b.profile(True)
            out[name] = b
        return out
    sptr = None if sim else C_void(torch.cuda.current_stream(dev).cuda_stream)
    # --two-streams: a stream per batch class, each class's exchange, verification and merge on ITS
    # stream.  The two batches' kernels then share the chip; both are bound by the SIMDs' issue
    # slots, so side by side they take 41.3 ms where one after the other takes 42.0 (r06v) — not the
    # default: the per-stage times of the line would overlap.
    two = not sim and args.two_streams
    streams = {name: (torch.cuda.Stream(device=dev) if two else None) for name in ("and", "phrase")}
    sptrs = {name: (C_void(streams[name].cuda_stream) if two else sptr) for name in streams}

    def on(name):
        return torch.cuda.stream(streams[name]) if two else contextlib.nullcontext()
    # the collective goes through the library's own RCCL communicator (irs_hip_comm_*), as in
    # the headline config; one communicator serves both exchanges
    comm = None
    if world > 1:
        comm = distributed.agreed_communicator(L, local_rank, rank, world, dev, log)
    ex = {name: distributed.PipelinedExchange(L, local_rank, n_segments, rank, world, nq, k, dev,
                                              comm=comm)
          for name in ("and", "phrase")}
    it = {"n": 0, "retire": []}
    kms = None

    def deliver(prev):
        # a finished step: verified (re-run if an estimate fell short), its per-segment lists
        # copied into the exchange slots, the all-gathers started
        bat, ph = prev
        for name, b in bat.items():
            with on(name):
                ex[name].finish(sptrs[name])
                hp, cp = ex[name].slot(ph, 0)
                b.results_to_device(hp, cp, sptrs[name])
                ex[name].start(ph)
        if rank == 0 and kms is not None:
            kms.append({n: b.timings() for n, b in bat.items()})
        # (destroyed one step later: irs_hip_batch_destroy waits for the copies just queued)
        for b in it["retire"]:
            it["reruns"] = it.get("reruns", 0) + b.reruns()
            b.close()
        it["retire"] = list(bat.values())

    def step():
        # software-pipelined one deep, as the headline config: the batches of step i are created
        # and their kernels queued FIRST, then step i-1 is verified and delivered — the host builds
        # step i while the GPU runs step i-1
        ph = it["n"] & 1
        bat = make_batches(it["n"])
        it["n"] += 1
        for name, b in bat.items():
            b.run(sptrs[name])
        prev = it.get("prev")
        if prev is not None:
            deliver(prev)
        it["prev"] = (bat, ph)

    def flush():
        prev = it.pop("prev", None)
        if prev is not None:
            deliver(prev)
        for name, e in ex.items():
            with on(name):
                e.finish(sptrs[name])
        sync()

    for _ in range(max(1, args.warmup)):
        step()
    flush()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    kms = []
    reruns_warm = it.get("reruns", 0) + sum(b.reruns() for b in it["retire"])
    marks = []
    for _ in range(args.steps):
        step()
        marks.append(time.perf_counter())   # (host clock after each step's submissions: where a stall sits)
    flush()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    marks.append(time.perf_counter())
    step_marks_ms = [round(1e3 * (b - a), 2) for a, b in zip([t0] + marks[:-1], marks)]
    for b in it["retire"]:
        it["reruns"] = it.get("reruns", 0) + b.reruns()
        b.close()
    it["retire"] = []
    reruns_timed = it.get("reruns", 0) - reruns_warm
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    bat = make_batches(0)
    alg = {n: b.work() for n, b in bat.items()}
    # what the kernels really decoded: one extra, untimed run with the counters on
    touched = {}
    for n, b in bat.items():
        b.profile(3)
        b.run(sptr)
        b.results()
        touched[n] = b.touched()
    vals = [alg["and"][0], alg["phrase"][0], touched["and"][0], touched["phrase"][0],
            touched["phrase"][1]]
    if world > 1:
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        vals = [float(x) for x in t.tolist()]
    if rank == 0:
        ms = {n: float(np.mean([x[n][_lib.K_SCORE] for x in kms])) for n in bat}
        stages = {n: [round(float(v), 4) for v in np.mean([x[n] for x in kms], axis=0)] for n in bat}
        tr5 = measured_traffic("traffic_config5_latest.json") if world == 1 else None
        rank0_touched = touched["and"][0] + touched["phrase"][0]
        rank0_alg = alg["and"][0] + alg["phrase"][0]
        out = {
            "metric": "queries/sec AND + by_phrase (positions), block-max WAND, TF-IDF, 50M-doc index @N GPU",
            "value": round(args.steps * 2 * nq / elapsed, 2), "unit": "queries/s", "n_gpus": world,
            "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u32+f32",
            "data": "synthetic" if not sim else "synthetic (EMULATOR DRY RUN)",
            "config": {
                "workload": "config 5: %d AND-of-2..4 + %d 2-word by_phrase queries/step, TF-IDF, "
                            "top-%d, %d-doc Zipfian index with positions, %d segments, WAND on the "
                            "AND batch" % (nq, nq, k, docs, n_segments),
                "segments": n_segments, "queries_per_step": 2 * nq,
                "algorithmic_bytes_per_step": {"and": int(vals[0]), "phrase_doc_side": int(vals[1])},
                "bytes_touched_per_step": {"and_doc_and_norm": int(vals[2]),
                                           "phrase_doc": int(vals[3]),
                                           "phrase_positions_read": int(vals[4])},
                "batches": "every step creates, runs and destroys its two batches (%d query sets in "
                           "rotation); steps are pipelined one deep" % n_sets,
                # (a batch whose candidates overflowed is executed twice: scores that tie by the
                # thousand at the threshold; the segments remember the slots that took)
                "reruns_rank0": int(it.get("reruns", 0)), "reruns_in_timed_steps": int(reruns_timed),
                "parallelism": "%d segments over %d GPU(s), all-gather of per-segment top-k + GPU merge"
                               % (n_segments, world),
                "collective": None if world == 1 else
                              ("irs_hip_topk_allgather (RCCL behind the C ABI)" if comm is not None
                               else "torch.distributed all_gather_into_tensor"),
                "rccl_library": None if comm is None else comm.library(),
                "ranks_seen": None if comm is None else comm.ranks_seen},
            "roofline": {"bound": "hbm", "kernel": "k_conj + k_phrase (rank 0)",
                         "achieved": round(rank0_touched / ((ms["and"] + ms["phrase"]) * 1e-3) / 1e9, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(rank0_touched / ((ms["and"] + ms["phrase"]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "peak_measured": HBM_MEASURED_GBS,
                         "frac_of_measured": round(rank0_touched / ((ms["and"] + ms["phrase"]) * 1e-3) / 1e9 / HBM_MEASURED_GBS, 5),
                         # (name and launch time only: the AND stage's bytes are not its)
                         "dominant_kernel": {"name": "k_phrase2" if ms["phrase"] >= ms["and"] else "k_conj",
                                             "stage_ms": round(max(ms["phrase"], ms["and"]), 4)},
                         # HBM-side bytes per step of those kernels (PMC passes of this command at these
                         # kernel sources, profiles/traffic_config5_latest.json), else null
                         "traffic": None if tr5 is None else int(tr5["bytes"]), "traffic_detail": tr5,
                         "kernel_ms": {"k_conj": round(ms["and"], 4),
                                                        "k_phrase": round(ms["phrase"], 4)},
                         # the same kernel time priced on A(q): every byte of the queries' lists
                         # (what the reference's iterators would at most decode)
                         "on_algorithmic_bytes": {
                             "achieved": round(rank0_alg / ((ms["and"] + ms["phrase"]) * 1e-3) / 1e9, 2),
                             "frac": round(rank0_alg / ((ms["and"] + ms["phrase"]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                         "stage_ms": {"order": ["plan", "pilot", "score", "select"], **stages},
                         "note": "achieved = bytes actually decoded / kernel time (block-driven "
                                 "kernels touch less than A(q))"},
            "cpu_baseline": None,
        }
        out["config"]["host_ms_between_steps"] = step_marks_ms   # (the last entry: flush)
        if world == 1 and not sim and not args.no_cpu:
            for b in bat.values():
                b.close()
            out["cpu_baseline"] = cpu_baseline_config5([segs[s] for s in range(n_segments)], ands,
                                                       phrases, sc, k)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main_tasks(args):
    """The reference's own task classes (index-search.cpp:91-449 over
    scripts/iresearch-benchmark.tasks), each as a batch of --queries distinct queries of the class
    — the task's words mapped to the Zipf ranks of the same document-frequency share, jittered
    +-25 % per query — executed (a) through the C ABI: every step a new batch, created, run,
    results delivered to host memory, destroyed, pipelined one deep; (b) by the restated CPU loop
    (oracle: index-search's heap loop) on the container's CPUs; with a parity check of the first
    queries of every class.  One line per category."""
    import concurrent.futures as cf

    import torch

    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity
    from iresearch_amd import _lib, search, synth, tasks
    from iresearch_amd.search import BM25, TFIDF
    sim = os.environ.get("IRS_BENCH_SIM")
    if sim:
        import ctypes
        L = _lib.bind(ctypes.CDLL(sim))
        sync = lambda: None
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        torch.cuda.set_device(0)
        L = _lib.lib()
        sync = torch.cuda.synchronize
    max_rank = args.max_rank if args.max_rank != 4096 else 262144
    nq = args.queries if args.queries != 1000 else 256
    k = args.k if args.k != 1000 else 100          # scripts/search-benchmark.sh: --topN=100
    t0 = time.perf_counter()
    seg = synth.build_segment(args.docs, max_rank, with_positions=True)
    log("built %d docs, ranks 1..%d with positions in %.1f s" % (args.docs, max_rank, time.perf_counter() - t0))
    sr = search.SegmentReader.from_synth(seg, device=0, L=L)
    st = [parity.segment_stats(seg)]
    scorer = BM25() if args.scorer == "bm25" else TFIDF(False)
    osc = parity.oracle_scorer(scorer)
    view = parity.oracle_view(seg)
    if args.tasks_file:
        with open(args.tasks_file) as f:
            lines = f.read().splitlines()
    else:   # the reference's own list, as tests/golden/make_tasks_golden.py extracted it
        with open(os.path.join(ROOT, "tests", "golden", "benchmark_tasks.json")) as f:
            lines = json.load(f)["lines"]
    cores = usable_cpus()
    native = False if sim else oracle.use_native()
    rows = []
    print("# %s, %d docs, ranks 1..%d, top-%d, %d distinct queries per class and step, CPU: %d threads (%s)" % (
        args.scorer, args.docs, max_rank, k, nq, cores, "-O3 -march=native" if native else "-O2"))
    print("# category            terms  ranks of the task's words          hits/query   GPU q/s    CPU q/s   GPU/CPU  parity"
          "           path   ms/step  plan/pilot/score/select ms  postings/step  re-runs (profiled batch + timed steps)")
    for t in tasks.parse_tasks(lines, 1):
        if t.category in tasks.UNION:
            # by_prefix / by_wildcard as the harness builds them (index-search.cpp:363-399:
            # scored_terms_limit; scripts/search-benchmark.sh: --scored-terms-limit=16): the visited
            # terms' `limit` longest posting lists scored as a disjunction, the rest as ONE unscored
            # bitset (limited_sample_collector.hpp, MultiTermQuery::execute multiterm_query.cpp
            # :112-184).  Here: one Or batch over the step's filters + irs_hip_bit_union_counts for
            # the totals (search.execute_expansions); the CPU leg: the oracle's harness loop over
            # the scored terms + its bit_union over all visited ones.
            limit = 16
            rng = np.random.default_rng(20260926 + len(rows))
            visits = [[tasks.expansion_of(t, max_rank, rng)] for _ in range(nq)]
            visits = [[v[0][np.asarray(seg.metas["docs_count"])[v[0]] > 0]] for v in visits]
            n_words = (seg.num_docs + 64) // 64
            n_par = min(nq, 8)
            prep = search.prepare_expansions(visits, limit, scorer, st)
            hits, counts, totals = search.execute_expansions([sr], prep, k)
            parity.check_expansions([seg], visits[:n_par], limit, scorer, k, hits[:, :n_par], counts[:, :n_par],
                                    totals[:, :n_par])
            steps = max(2, args.steps)
            host = [0.0, 0.0]
            for timed in (False, True):
                sync()
                t0 = time.perf_counter()
                for _ in range(steps if timed else 2):
                    ta = time.perf_counter()
                    prep = search.prepare_expansions(visits, limit, scorer, st)
                    tb = time.perf_counter()
                    search.execute_expansions([sr], prep, k)
                    if timed:
                        host[0] += tb - ta
                        host[1] += time.perf_counter() - tb
                sync()
                gpu_dt = (time.perf_counter() - t0) / (steps if timed else 2)
            metas_all = np.zeros(len(seg.metas), oracle.TERM_META)
            for name in oracle.TERM_META.names:
                metas_all[name] = seg.metas[name]

            def cpu_one(i):
                p = prep[i]
                if p.scored:
                    oracle.search([view], metas_all[np.array(p.scored)][None], oracle.OP_OR, osc, k, None)
                return oracle.bit_union(seg.doc_file, metas_all[visits[i][0]], seg.layout, True, n_words)
            t0 = time.perf_counter()
            with cf.ThreadPoolExecutor(cores) as ex:
                list(ex.map(cpu_one, range(nq)))
            cpu_dt = time.perf_counter() - t0
            row = {"category": t.category, "terms": int(np.mean([len(v[0]) for v in visits])),
                   "form": "scored_terms_limit=%d: a disjunction of the %d longest lists + one unscored "
                           "bit_union (counts only)" % (limit, limit),
                   "hits_per_query": float(totals.mean()), "gpu_qps": nq / gpu_dt, "cpu_qps": nq / cpu_dt,
                   "cpu_sample": nq, "parity_checked": n_par, "ms_per_step": gpu_dt * 1e3,
                   "path": "joined + bit_union_counts",
                   "host_ms": {"prepare": round(1e3 * host[0] / steps, 3), "execute": round(1e3 * host[1] / steps, 3)}}
            rows.append(row)
            print("%-20s %5d  %-34s %10.0f %10.0f %10.1f %8.1fx  ok (%d queries)  %-6s %7.3f  (scored_terms_limit=%d; host: prepare %.2f + execute %.2f ms)" % (
                t.category, row["terms"], "'%s' -> terms visited" % t.text.strip(), row["hits_per_query"],
                row["gpu_qps"], row["cpu_qps"], row["gpu_qps"] / row["cpu_qps"], n_par, "expand", row["ms_per_step"],
                limit, row["host_ms"]["prepare"], row["host_ms"]["execute"]), flush=True)
            continue
        if t.category in tasks.EXPANSION:
            print("%-20s (multi-term expansion filter: not on this path)" % t.category)
            continue
        rng = np.random.default_rng(20260926 + len(rows))
        base = tasks.ranks_of(t, max_rank)
        queries = [tasks.ranks_of(t, max_rank, 0.25, rng) for _ in range(nq)]
        filters = [tasks.filter_of(t, r) for r in queries]
        phrase = t.category in tasks.PHRASE

        def make():   # (the filters are prepared inside the step, as the harness does)
            return sr.batch(search.prepare_filters(filters, scorer, st, [sr], k), k).run()
        # parity of the first queries of the class
        b = make()
        hits, counts, totals = (x.copy() for x in b.results_to_host().host_results())
        b.close()
        n_par = min(nq, 32)
        check = parity.check_phrase_segment if phrase else parity.check_single_segment
        check(seg, filters[:n_par], scorer, k, hits[:n_par], counts[:n_par], totals[:n_par])
        # where a step of the class goes: one more batch with the kernels timed (HIP events on the
        # batch's stream), the path it took, re-runs, postings and the host stages in front of it
        t0 = time.perf_counter()
        prep = search.prepare_filters(filters, scorer, st, [sr], k)
        t1 = time.perf_counter()
        b = sr.batch(prep, k)
        t2 = time.perf_counter()
        b.profile(1).run()
        b.results_to_host().host_results()
        t3 = time.perf_counter()
        stage_ms = dict(zip(("plan", "pilot", "score", "select"), b.timings()))
        diag = {"path": ({_lib.PATH_ITEMS: "items", _lib.PATH_JOINED: "joined"}.get(b.path(), "?")
                         + ("+pairs" if b.paired_tiles() else "")) if not phrase else "phrase",
                "reruns": b.reruns(), "postings_per_step": b.work()[1],
                "stage_ms": {n: round(v, 3) for n, v in stage_ms.items()},
                "host_ms": {"prepare": round((t1 - t0) * 1e3, 3), "create": round((t2 - t1) * 1e3, 3),
                            "run_to_results": round((t3 - t2) * 1e3, 3)}}
        b.close()
        # GPU: fresh batches, one deep
        steps = max(2, args.steps)
        for timed in (False, True):
            sync()
            t0 = time.perf_counter()
            prev = None
            reruns = 0
            # (untimed first: three pipelined steps, so that the pool holds the blocks of two
            # batches of this class — a first-use hipMalloc / hipHostMalloc costs more than a step)
            for _ in range(steps if timed else 3):
                cur = make()
                if prev is not None:
                    prev.results_to_host().host_results()
                    reruns += prev.reruns()
                    prev.close()
                prev = cur
            prev.results_to_host().host_results()
            reruns += prev.reruns()
            prev.close()
            sync()
            gpu_dt = (time.perf_counter() - t0) / (steps if timed else 3)
        # CPU: the same queries (a bounded sample), the threads pop one task queue
        metas = [parity.metas_for(seg, [r - 1 for r in q])[None] for q in queries]

        def run_cpu(i):
            if phrase:
                oracle.search_phrase([view], metas[i], list(range(len(queries[i]))), osc, k, 1.0)
            else:
                flt = filters[i]
                op = parity.oracle_op(flt, search._terms_of(flt)[0])
                oracle.search([view], metas[i], op, osc, k, None)

        def timed_cpu(n):
            t0 = time.perf_counter()
            with cf.ThreadPoolExecutor(cores) as ex:
                list(ex.map(run_cpu, [i % nq for i in range(n)]))
            return time.perf_counter() - t0
        probe = min(nq, 2 * cores)
        dt = max(timed_cpu(probe), 1e-6)
        n_cpu = int(min(4 * nq, max(probe, probe * 3.0 / dt)))
        cpu_dt = timed_cpu(n_cpu)
        row = {"category": t.category, "terms": len(base), "ranks": base,
               "hits_per_query": float(totals.mean()), "gpu_qps": nq / gpu_dt,
               "cpu_qps": n_cpu / cpu_dt, "cpu_sample": n_cpu, "parity_checked": n_par,
               "ms_per_step": gpu_dt * 1e3, "reruns_in_timed_steps": reruns, **diag}
        rows.append(row)
        sm = diag["stage_ms"]
        print("%-20s %5d  %-34s %10.0f %10.0f %10.1f %8.0fx  ok (%d queries)  %-6s %7.3f  %5.2f/%5.2f/%5.2f/%5.2f  %6.1f M  %d+%d" % (
            t.category, len(base), str(base if len(base) <= 4 else base[:4] + ["..."]), row["hits_per_query"],
            row["gpu_qps"], row["cpu_qps"], row["gpu_qps"] / row["cpu_qps"], n_par, diag["path"],
            row["ms_per_step"], sm["plan"], sm["pilot"], sm["score"], sm["select"],
            diag["postings_per_step"] / 1e6, diag["reruns"], reruns), flush=True)
    print(json.dumps({"tasks": rows, "docs": args.docs, "max_rank": max_rank, "k": k, "queries_per_step": nq,
                      "scorer": args.scorer, "cpu_threads": cores,
                      "data": "synthetic" if not sim else "synthetic (EMULATOR DRY RUN)"}), flush=True)
    sr.close()


def kernel_sources_sha():
    """What a traffic measurement is valid for: the kernel sources it was taken with."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "iresearch_amd", "csrc")
    for n in sorted(os.listdir(d)) + sorted("hip/" + x for x in os.listdir(os.path.join(d, "hip"))):
        p = os.path.join(d, n)
        if os.path.isfile(p) and n.endswith((".h", ".hip")):
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(name="traffic_latest.json"):
    """HBM bytes per k_score launch from rocprofv3 PMC passes of this same command
    (FETCH_SIZE and WRITE_SIZE in their own --pmc runs, tools/summarize_prof.py), committed
    under profiles/ — reported only while the kernel sources are the ones it was measured
    with; null otherwise (a stale figure is worse than none)."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    return t if t.get("kernel_sources_sha") == kernel_sources_sha() else None


def C_void(v):
    import ctypes
    return ctypes.c_void_p(v)


if __name__ == "__main__":
    main()
