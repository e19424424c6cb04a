#!/usr/bin/env python3
"""Where does tri_batch_create's time go INSIDE the serving loop?  cfg2, 16384 queries: the create timed (a) back to back on the main thread,
(b) on a compiler thread while the main thread sleeps, (c) while the main thread runs / awaits / reads back a resident batch over and over (what
bench.py's Pipeline does), (d) as (c) with the main thread pinned to the far end of the affinity mask.  Prints the median wall time per create and the
engine's own create_ms / create_plan_ms.    python tools/probe_create_loop.py [plan_threads]"""
import os, sys, time, threading, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trinity_amd as T
from trinity_amd import workloads as W

parts, desc = W.build_parts("cfg2", 10_000_000, 1_000_000, 10, 42, 16384)
pt = parts[0]
seg = T.Segment(10_000_000, 1_000_000, 10, 42, codec=pt.codec)
dev = T.Device(0)
if len(sys.argv) > 1:
    dev.set_option("plan_threads", int(sys.argv[1]))
ix = T.Index.from_segment(dev, seg)
flat = T.engine.flatten(pt.programs)
mk = lambda: T.Batch(ix, None, pt.flags, topk=pt.topk, flat=flat)
res = mk(); res.run(); res.sync(); res.counts()
for _ in range(4):  # (pool buffers warm)
    b = mk(); b.close()

def creates(n, out):
    for _ in range(n):
        t0 = time.perf_counter(); b = mk(); dt = (time.perf_counter() - t0) * 1e3
        i = b.info(); out.append((dt, i["create_ms"], i["create_plan_ms"])); b.close()

def report(tag, out):
    out = out[3:]
    print(f"{tag:60s} wall {statistics.median(o[0] for o in out):.3f}  engine create_ms {statistics.median(o[1] for o in out):.3f}  plan_ms {statistics.median(o[2] for o in out):.3f}", flush=True)

out = []; creates(40, out); report("(a) main thread, back to back", out)
out = []; th = threading.Thread(target=creates, args=(40, out)); th.start(); th.join(); report("(b) compiler thread, main thread joins (sleeps)", out)
for pin in (False, True):
    if pin:
        cpus = sorted(os.sched_getaffinity(0)); os.sched_setaffinity(0, {cpus[len(cpus) // 2 - 1]})  # (the calling thread only)
    out = []; stop = [False]
    th = threading.Thread(target=lambda: (creates(60, out), stop.__setitem__(0, True))); th.start()
    steps = 0; t0 = time.perf_counter()
    while not stop[0]:
        res.run(); res.sync(); res.counts(); steps += 1
    dt = time.perf_counter() - t0
    th.join(); report(f"({'d' if pin else 'c'}) compiler thread while main runs/syncs{' (main pinned far)' if pin else ''}: {dt / max(1, steps) * 1e3:.3f} ms/step", out)
