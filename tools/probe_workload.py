#!/usr/bin/env python3
"""Perf probe (GPU): run one SURVEY §8(d) workload and print step time (+ per-phase cycle shares on TRI_PROF builds).
   WORKLOAD=cfg3 NQ=2048 [TRINITY_HIP_LIB=build/libtrinity_hip_prof.so] python tools/probe_workload.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("TRINITY_HIP_LIB"):
    import trinity_amd.engine as E
    E.LIB_HIP = os.path.abspath(os.environ["TRINITY_HIP_LIB"])
import trinity_amd as T
from trinity_amd import workloads as W
import trinity_amd.engine as E

name = os.environ.get("WORKLOAD", "cfg3")
D, V, NQ = int(os.environ.get("DOCS", 10_000_000)), int(os.environ.get("VOCAB", 1_000_000)), int(os.environ.get("NQ", 2048))
if name == "or5":  # cfg5's pure 5-way unions alone (DocumentsOnly, google_codec)
    progs, flags, topk, codec, desc = W.or5(E.gen_queries(V, 1337 + 2, NQ, 5)), E.FLAG_DOCUMENTS_ONLY, 0, E.CODEC_GOOGLE, "or5: 5-way OR, google_codec, DocumentsOnly"
else:
    progs, flags, topk, codec, desc = W.build(name, D, V, 10, 42, NQ)  # single-part workloads (cfg5 runs as two batches: bench.py)
if os.environ.get("ONLY"):  # cfg3's query classes alone: ONLY=0 `A B (C|D|E)`, 1 `(A|B) (C|D) E`, 2 `A B C D E`, 3 `A|B|C|D|E`
    keep = int(os.environ["ONLY"])
    progs = [p for i, p in enumerate(progs) if (i & 3) == keep]
    desc += f" [class {keep} only: {len(progs)} queries]"
if os.environ.get("CODEC"):
    codec = int(os.environ["CODEC"])
    desc += f" [codec forced to {codec}]"
seg = T.Segment(D, V, 10, 42, codec=codec)
dev = T.Device(0)
for kv in filter(None, os.environ.get("OPTIONS", "").split(",")):  # OPTIONS="planes=0,fused_task_cost=1048576"
    dev.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ix = T.Index.from_segment(dev, seg)
if os.environ.get("RICH"):
    flags, topk = T.FLAG_MATCHED_TERMS, 0
    desc += " [default (rich match) mode]"
if os.environ.get("EACH"):  # EACH=n: the first n queries one batch each — the per-query time distribution (a kernel's tail is its longest task)
    rows = []
    for p in progs[: int(os.environ["EACH"])]:
        b1 = T.Batch(ix, [p], flags, topk=topk)
        for _ in range(2):
            b1.run(); b1.sync()
        i1 = b1.info()
        rows.append((i1["last_run_ms"], [int(x) for x in p], int(i1["matches"])))
        b1.close()
    if os.environ.get("EACH_OUT"):  # every row, for fitting the planner's cost model offline
        import json
        json.dump([{"ms": r[0], "terms": [x & 0x0FFFFFFF for x in r[1] if x >> 28 == E.OP_TERM], "matches": r[2], "df": [int(seg.terms[x & 0x0FFFFFFF][0]) if hasattr(seg, "terms") else 0 for x in r[1] if x >> 28 == E.OP_TERM]} for r in rows], open(os.environ["EACH_OUT"], "w"))
    rows.sort(key=lambda r: -r[0])
    ms = [r[0] for r in rows]
    print(f"  per query: mean {sum(ms) / len(ms):.3f} ms  max {ms[0]:.3f}  median {ms[len(ms) // 2]:.3f}  p90 {ms[len(ms) // 10]:.3f}")
    for r in rows[:12] + rows[-3:]:
        print(f"    {r[0]:.3f} ms  matches {r[2]:>9}  terms {[x & 0x0FFFFFFF for x in r[1] if x >> 28 == E.OP_TERM]}")
b = T.Batch(ix, progs, flags, topk=topk)
best = 1e9
for _ in range(int(os.environ.get('RUNS', 3))):
    import time as _t
    _t0 = _t.perf_counter(); b.run(); b.sync(); wall = (_t.perf_counter() - _t0) * 1e3
    best = min(best, wall if os.environ.get("RICH") else b.info()["last_run_ms"])  # rich mode: sync runs the WRITE pass too
inf = b.info()
L = E.hip_lib()
if hasattr(L, "tri_debug_prof"):
    buf = (C.c_uint64 * 32)(); L.tri_debug_prof(buf); v = list(buf)[:16]; tot = sum(v) or 1
    print("  prof " + " ".join(f"p{i}={x / tot * 100:.1f}%" for i, x in enumerate(v) if x), f"(total {tot:.3e} cycles)")
    c = list(buf)[16:28]
    w = list(buf)[28:32]
    if w[0]:  # k_planes workgroups: when they ended relative to the first one's start (100 MHz clock), last run(s)
        t0 = (~w[1]) & (2**64 - 1)
        print(f"  workgroups {w[0]}: mean end {(w[2] / w[0] - t0) / 100:.1f} us, last end {(w[3] - t0) / 100:.1f} us after the first start")
    if any(c):  # event counters (k_planes: 16 candidate steps, 17 frequency lookups, 18 prunes, 19 sub-windows, 20 table-lookup steps, 23 sweeps cut short by a full queue / buffer), over the 3 runs
        print("  counters " + " ".join(f"c{16 + i}={x}" for i, x in enumerate(c) if x))
print("  " + " ".join(f"{k}={inf[k]:.3f}" for k in ("term_planes_ms", "dense_ms", "pset_ms", "probe_ms", "cand_ms", "fused_ms", "planes_ms", "phrase_ms", "rest_ms")), f"fused_q={inf['fused_queries']} planes_q={inf['planes_queries']} cand_q={inf['cand_queries']} plane_terms={inf['plane_terms']}")
print(f"{desc}: {len(progs)} queries {best:.2f} ms  matches {inf['matches']:.3e}  alg {inf['algorithmic_bytes'] / best / 1e6:.1f} GB/s  {NQ / best * 1e3:.0f} q/s")
