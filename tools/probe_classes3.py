#!/usr/bin/env python3
"""Perf probe (GPU): cfg3 as a whole and each of its four query classes alone, one process (one segment build).
   On -DTRI_PROF builds prints k_planes' per-phase cycle shares, on -DTRI_PROF_COUNTS builds its event counters.
   [TRINITY_HIP_LIB=build/libtrinity_hip_prof.so] [NQ=8192] [OPTIONS=k=v,...] python tools/probe_classes3.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trinity_amd.engine as E
if os.environ.get("TRINITY_HIP_LIB"):
    E.LIB_HIP = os.path.abspath(os.environ["TRINITY_HIP_LIB"])
import trinity_amd as T
from trinity_amd import workloads as W

name = os.environ.get("WORKLOAD", "cfg3")
D, V, NQ = int(os.environ.get("DOCS", 10_000_000)), int(os.environ.get("VOCAB", 1_000_000)), int(os.environ.get("NQ", 8192))
progs, flags, topk, codec, desc = W.build(name, D, V, 10, 42, NQ)
seg = T.Segment(D, V, 10, 42, codec=codec)
dev = T.Device(0)
for kv in filter(None, os.environ.get("OPTIONS", "").split(",")):
    dev.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ix = T.Index.from_segment(dev, seg)
L = E.hip_lib()
names = {None: "all", 0: "A B (C|D|E)", 1: "(A|B) (C|D) E", 2: "A B C D E", 3: "A|B|C|D|E"}
for keep in [None if c == 'all' else int(c) for c in os.environ.get('CLASSES', 'all,0,1,2,3').split(',')]:
    ps = progs if keep is None else [p for i, p in enumerate(progs) if (i & 3) == keep]
    b = T.Batch(ix, ps, flags, topk=topk)
    b.run(); b.sync()  # (planes built, pools warm)
    if hasattr(L, "tri_debug_prof"):
        buf = (C.c_uint64 * 32)(); L.tri_debug_prof(buf)  # reset
    best = 1e9
    R = int(os.environ.get("RUNS", 3))
    for _ in range(R):
        b.run(); b.sync()
        best = min(best, b.info()["last_run_ms"])
    inf = b.info()
    print(f"[{names[keep]}] {len(ps)} queries: {best:.2f} ms  " + " ".join(f"{k}={inf[k]:.3f}" for k in ("cand_ms", "fused_ms", "planes_ms", "rest_ms")), f"planes_q={inf['planes_queries']} cand_q={inf['cand_queries']} fused_q={inf['fused_queries']} matches={inf['matches']:.3e}")
    if hasattr(L, "tri_debug_prof"):
        buf = (C.c_uint64 * 32)(); L.tri_debug_prof(buf); v = list(buf)[:16]; tot = sum(v) or 1
        print("    prof " + " ".join(f"p{i}={x / tot * 100:.1f}%" for i, x in enumerate(v) if x), f"(total {tot:.3e} cycles over {R} runs)")
        c = list(buf)[16:28]
        if any(c):
            print("    counters/run " + " ".join(f"c{16 + i}={x // R}" for i, x in enumerate(c) if x))
    b.close()
