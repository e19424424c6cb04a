#!/usr/bin/env python3
"""tri_batch_create's host planner timed WITHOUT a device (csrc/planner.hpp through libtrinity_host.so):
    WORKLOAD=cfg2 NQ=16384 THREADS=1,2,4,8 DOCS=10000000 VOCAB=1000000 python tools/plan_probe.py
prints, per thread count, the best of RUNS plans and the four phase times (lowering + classes, tasks, layout + fill, schedule)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import trinity_amd as T
from trinity_amd import hostplan as HP
from trinity_amd import workloads as W

from trinity_amd import build as _B
_B.build_host()
wl = os.environ.get("WORKLOAD", "cfg2")
docs, vocab = int(os.environ.get("DOCS", 10_000_000)), int(os.environ.get("VOCAB", 1_000_000))
nq = int(os.environ.get("NQ", {"cfg3": 8192, "cfg5": 12500}.get(wl, 16384)))
runs = int(os.environ.get("RUNS", 7))
parts, desc = W.build_parts(wl, docs, vocab, 10, 42, nq)
print(desc, flush=True)
segs = {}
for pt in parts:
    if pt.codec not in segs:
        t0 = time.time()
        seg = T.Segment(docs, vocab, 10, 42, codec=pt.codec)
        t1 = time.time()
        segs[pt.codec] = (seg, HP.HostIndex.from_segment(seg))
        print(f"segment codec {pt.codec}: build {t1 - t0:.1f}s, host index {time.time() - t1:.1f}s", flush=True)
opts = dict(kv.split("=") for kv in filter(None, os.environ.get("OPTIONS", "").split(",")))
opts = {k: int(v) for k, v in opts.items()}
for pt in parts:
    flat = HP.flatten(pt.programs)
    for th in [int(x) for x in os.environ.get("THREADS", "1,2,4,8").split(",")]:
        best = None
        for _ in range(runs):
            t0 = time.perf_counter()
            p = HP.HostPlan(segs[pt.codec][1], None, pt.flags, pt.topk, threads=th, options=opts, flat=flat)
            dt = (time.perf_counter() - t0) * 1e3
            if best is None or p.ms.sum() < best[1].sum():
                best = (dt, p.ms.copy(), dict(p.s))
            p.close()
        s = best[2]
        print(f"{pt.name[:24]:24s} threads {th:2d}: plan {best[1].sum():7.3f} ms  (lower {best[1][0]:.3f}  tasks {best[1][1]:.3f}  fill {best[1][2]:.3f}  sched {best[1][3]:.3f})  call {best[0]:.2f} ms (incl. pool start)  "
              f"queries {s['n_plan']} tasks {s['n_tasks']} dense/cand/fused/planes {s['dense_queries']}/{s['cand_queries']}/{s['fused_queries']}/{s['planes_queries']} plane_terms {s['n_plane_terms']} block {s['block_bytes'] >> 10} KB", flush=True)
