import time, numpy as np, torch
from iresearch_amd import _lib, search, synth
from iresearch_amd.search import BM25, Or, by_term
seg = synth.build_segment(2_000_000, 4096)
sr = search.SegmentReader.from_synth(seg)
st = search.SegmentStats(seg.docs_with_field, seg.total_term_freq, np.asarray(seg.metas["docs_count"]))
for rep in range(3):
    ranks = synth.make_queries(1000, 8, 16, 4096, synth.SEED + 10 + rep)
    prep = search.prepare([Or([by_term(int(r) - 1) for r in row]) for row in ranks], BM25(), [st])
    t0 = time.perf_counter(); b = sr.batch(prep, 1000); t1 = time.perf_counter()
    b.run(); torch.cuda.synchronize(); t2 = time.perf_counter()
    hits, counts, totals = b.results(); t3 = time.perf_counter()
    b.run(); torch.cuda.synchronize(); t4 = time.perf_counter()
    b.results(); t5 = time.perf_counter()
    print("create %.1f ms, first run %.1f, first results %.1f, second run %.1f, second results %.1f" % (
        1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3), 1e3*(t5-t4)), flush=True)
    b.close()
