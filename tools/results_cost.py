#!/usr/bin/env python3
"""Dev utility: what handing the top-k back to the HOST costs per step (the PCIe-inclusive
rate of DESIGN.md §3.2): the headline batch with device-resident results vs
irs_hip_batch_results."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iresearch_amd import search, synth
from iresearch_amd.search import BM25, Or, by_term
seg = synth.build_segment(10_000_000, 4096)
sr = search.SegmentReader.from_synth(seg)
ranks = synth.make_queries(1000, 8, 16, 4096, synth.SEED + 2)
filters = [Or([by_term(int(r) - 1) for r in row]) for row in ranks]
st = search.SegmentStats(seg.docs_with_field, seg.total_term_freq, np.asarray(seg.metas["docs_count"]))
prep = search.prepare(filters, BM25(), [st])
b = sr.batch(prep, 1000)
b.run(); b.results()
for name, fn in (("run + device results", lambda: (b.run(), b.device_results())),
                 ("run + results to host", lambda: (b.run(), b.results()))):
    fn()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    dt = (time.perf_counter() - t0) / 5
    print("%s: %.2f ms/step = %.0f queries/s" % (name, dt * 1e3, 1000 / dt), flush=True)
