#!/usr/bin/env python3
"""Perf probe (GPU): split the cfg2 batch by the planner's execution class and time each class alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("TRINITY_HIP_LIB"):
    import trinity_amd.engine as E
    E.LIB_HIP = os.path.abspath(os.environ["TRINITY_HIP_LIB"])
import trinity_amd as T

D, V, NQ = 10_000_000, 1_000_000, int(os.environ.get("NQ", 16384))
seg = T.Segment(D, V, 10, 42)
dev = T.Device(0)
ix = T.Index.from_segment(dev, seg)
df = seg.terms[:, 0].astype(np.int64)
nb = (df + 31) // 32
qs = T.gen_queries(V, 1337, NQ, 2)
d = df[qs]
lead = d.min(1); other = d.max(1); nbo = (other + 31) // 32
bd = nbo <= lead
dense = bd & (d.sum(1) >= 512 * 1024)
classes = {"dense(bitmap windows)": dense, "cand block-driven": bd & ~dense, "cand galloping": ~bd}

def prof_dump(tag):
    """TRI_PROF builds (TRINITY_HIP_LIB=...): per-phase cycle totals of wave 0 of every workgroup."""
    import ctypes as C
    L = E_lib()
    if not hasattr(L, "tri_debug_prof"):
        return
    buf = (C.c_uint64 * 32)()
    L.tri_debug_prof(buf)
    v = list(buf)[:16]
    tot = sum(v) or 1
    print(f"  prof[{tag}] " + " ".join(f"p{i}={x / tot * 100:.1f}%" for i, x in enumerate(v) if x), f"(total {tot:.3e} cycles)")


def E_lib():
    import trinity_amd.engine as E
    return E.hip_lib()


def run(q, reps=3):
    b = T.Batch.conjunctions(ix, q)
    best = 1e9
    for _ in range(reps):
        b.run(); b.sync(); best = min(best, b.info()["last_run_ms"])
    inf = b.info(); b.close()
    prof_dump("%d queries" % len(q))
    return best, inf

only = os.environ.get("CLASS")
if only:
    classes = {k: v for k, v in classes.items() if k.startswith(only)}
tot = 0
for name, m in classes.items():
    q = qs[m]
    if not len(q):
        continue
    ms, inf = run(q)
    tot += ms
    print(f"{name:24s} n={len(q):6d} {ms:8.3f} ms  lead docs {lead[m].sum():.3e} other postings {other[m].sum():.3e} matches {inf['matches']:.3e} alg {inf['algorithmic_bytes']/ms/1e6:9.1f} GB/s  ({(lead[m].sum()+other[m].sum())/ms/1e6:.1f} Mpostings/ms if fully decoded)", flush=True)
if only:
    sys.exit(0)
ms, inf = run(qs)
print(f"all {ms:.3f} ms (sum of classes {tot:.3f})  alg {inf['algorithmic_bytes']/ms/1e6:.1f} GB/s")
