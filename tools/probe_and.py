#!/usr/bin/env python3
"""Perf probe (GPU): kernel time of k_and for single queries and homogeneous batches of query classes on the
L segment — tells which regime (dense x dense, dense x sparse gallop, tiny) dominates the cfg2 batch."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import trinity_amd as T

D, V = int(os.environ.get("DOCS", 10_000_000)), int(os.environ.get("VOCAB", 1_000_000))
seg = T.Segment(D, V, 10, 42)
dev = T.Device(0)
ix = T.Index.from_segment(dev, seg)
df = seg.terms[:, 0]

def run(qs, reps=3):
    b = T.Batch.conjunctions(ix, np.array(qs, dtype=np.uint32))
    best = 1e9
    for _ in range(reps):
        b.run(); b.sync()
        best = min(best, b.info()["last_run_ms"])
    inf = b.info()
    b.close()
    return best, inf

print("== single queries (one workgroup)")
for q in [[0, 1], [0, 2], [1, 2], [0, 10], [0, 100], [0, 1000], [0, 10000], [0, 100000], [10, 20], [100, 200], [1000, 2000], [10000, 20000]]:
    ms, inf = run([q])
    post = int(df[q[0]]) + int(df[q[1]])
    print(f"q={q} df=({df[q[0]]},{df[q[1]]}) matches={inf['matches']} {ms:.3f} ms  {post/ms/1e6:.1f} Mpost/ms-equivalent  alg {inf['algorithmic_bytes']/ms/1e6:.1f} GB/s", flush=True)
print("== homogeneous batches of 2048 queries")
rng = np.random.default_rng(0)
for name, lo0, hi0, lo1, hi1 in [("head x head", 0, 8, 0, 8), ("head x mid", 0, 8, 100, 1000), ("head x rare", 0, 8, 10000, 100000), ("mid x mid", 100, 1000, 100, 1000), ("mid x rare", 100, 1000, 10000, 100000), ("rare x rare", 10000, 100000, 10000, 100000), ("tail x tail", 100000, 900000, 100000, 900000)]:
    a = rng.integers(lo0, hi0, 2048); b = rng.integers(lo1, hi1, 2048)
    b = np.where(a == b, b + 1, b)
    ms, inf = run(np.stack([a, b], 1).tolist())
    print(f"{name:14s} {ms:8.3f} ms  {2048/ms*1e3:12.0f} q/s  alg {inf['algorithmic_bytes']/ms/1e6:9.1f} GB/s  matches {inf['matches']}", flush=True)
print("== cfg2 batch sizes")
for n in (1024, 4096, 16384):
    qs = T.gen_queries(V, 1337, n, 2)
    ms, inf = run(qs.tolist())
    print(f"n={n} {ms:.3f} ms {n/ms*1e3:.0f} q/s alg {inf['algorithmic_bytes']/ms/1e6:.1f} GB/s", flush=True)
