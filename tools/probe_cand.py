#!/usr/bin/env python3
"""Perf probe (GPU): cfg2's candidate-tile queries (k_and) split by how their second list is tested — a bit probe in the term's plane
(the partner is a head term) or galloping / block-driven merges of its blocks — each class run alone; prints k_and's time per class."""
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
for kv in filter(None, os.environ.get("OPTIONS", "").split(",")):
    dev.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ix = T.Index.from_segment(dev, seg)
df = seg.terms[:, 0].astype(np.int64)
qs = T.gen_queries(V, 1337, NQ, 2)
d = df[qs]
lead, other = d.min(1), d.max(1)
plane = D // 128
classes = {"both planes": (lead >= plane), "probe (partner has a plane)": (lead < plane) & (other >= plane), "no plane: other blocks <= lead docs": (other < plane) & ((other + 31) // 32 <= lead),
           "no plane: galloping": (other < plane) & ((other + 31) // 32 > lead)}
for name, m in classes.items():
    q = qs[m]
    if not len(q):
        continue
    b = T.Batch.conjunctions(ix, q)
    best = None
    for _ in range(4):
        b.run(); b.sync()
        i = b.info()
        if best is None or i["last_run_ms"] < best["last_run_ms"]:
            best = i
    b.close()
    print(f"{name:40s} n={len(q):6d} step {best['last_run_ms']:.3f} ms  k_and {best['cand_ms']:.3f} k_psets {best['pset_ms']:.3f} k_probe {best['probe_ms']:.3f} k_and_dense {best['dense_ms']:.3f} planes {best['term_planes_ms']:.3f}  "
          f"lead docs {lead[m].sum():.3e} (mean {lead[m].mean():.0f}, median {np.median(lead[m]):.0f}) other {other[m].sum():.3e} matches {best['matches']:.3e} cand_q {best['cand_queries']}", flush=True)
    L = T.engine.hip_lib()
    if hasattr(L, "tri_debug_prof"):
        import ctypes as C
        buf = (C.c_uint64 * 32)(); L.tri_debug_prof(buf); v = list(buf)[:16]; tot = sum(v) or 1
        print("      prof " + " ".join(f"p{i}={x / tot * 100:.1f}%" for i, x in enumerate(v) if x), f"(total {tot:.3e} cycles)", flush=True)
