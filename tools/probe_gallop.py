#!/usr/bin/env python3
"""Perf probe (GPU): the galloping class of the cfg2 batch, binned by lead-list size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("TRINITY_HIP_LIB"):
    import trinity_amd.engine as E
    E.LIB_HIP = os.path.abspath(os.environ["TRINITY_HIP_LIB"])
import trinity_amd as T
D, V, NQ = 10_000_000, 1_000_000, 16384
seg = T.Segment(D, V, 10, 42); dev = T.Device(0); ix = T.Index.from_segment(dev, seg)
df = seg.terms[:, 0].astype(np.int64)
qs = T.gen_queries(V, 1337, NQ, 2); d = df[qs]; lead = d.min(1); other = d.max(1)
gal = ((other + 31) // 32) > lead
def run(q, reps=3):
    b = T.Batch.conjunctions(ix, q); best = 1e9
    for _ in range(reps):
        b.run(); b.sync(); best = min(best, b.info()["last_run_ms"])
    inf = b.info(); b.close(); return best, inf
for lo, hi in [(0, 64), (64, 1024), (1024, 8192), (8192, 32768), (32768, 1 << 30)]:
    m = gal & (lead >= lo) & (lead < hi)
    if not m.any(): continue
    ms, inf = run(qs[m])
    print(f"lead [{lo},{hi}) n={m.sum():5d} cands {lead[m].sum():.3e} {ms:7.3f} ms  {lead[m].sum()/ms/1e6:8.2f} Mcand/ms  {m.sum()/ms:8.1f} q/ms  tasks/q~{max(1,int(np.ceil(lead[m].mean()/8192)))}", flush=True)
