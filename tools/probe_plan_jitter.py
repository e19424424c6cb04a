#!/usr/bin/env python3
"""Do the planner's parallel passes stall without a device, without Python threads, without a second pool?  The host planner (libtrinity_host.so) called back to back
N times on cfg2's 16384 queries with 16 threads: the distribution of a call's wall time and of its four phase times.  python tools/probe_plan_jitter.py [N] [threads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import trinity_amd as T
from trinity_amd import hostplan as HP, workloads as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 16
parts, _ = W.build_parts("cfg2", 10_000_000, 1_000_000, 10, 42, 16384)
seg = T.Segment(10_000_000, 1_000_000, 10, 42, codec=1)
hi = HP.HostIndex.from_segment(seg)
flat = HP.flatten(parts[0].programs)
walls, phases = [], []
for i in range(n):
    t0 = time.perf_counter()
    p = HP.HostPlan(hi, None, parts[0].flags, 0, threads=thr, options={"frag_cache": 1}, flat=flat)
    walls.append((time.perf_counter() - t0) * 1e3)
    phases.append(p.ms.copy())
    p.close()
walls = np.array(walls[5:]); phases = np.array(phases[5:])
print(f"{len(walls)} plans on {thr} threads: wall ms p50 {np.median(walls):.3f} p99 {np.percentile(walls, 99):.3f} max {walls.max():.3f}; calls over 3 ms: {(walls > 3).sum()}")
print("phase max ms (lower, tasks, fill, sched):", np.round(phases.max(axis=0), 3), " p50:", np.round(np.median(phases, axis=0), 3))
