#!/usr/bin/env python3
"""Perf probe (GPU): the write side — re-encode the postings of the synthetic segment with tri_encode_google and time it.
   DOCS=10000000 VOCAB=1000000 python tools/probe_encode.py
Positions are not kept by the read side's bulk decode, so every document's hits get positions 1..freq (the byte volume of the
real corpus' positions 1..10 is the same: one byte per hit)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import trinity_amd as T

D, V = int(os.environ.get("DOCS", 10_000_000)), int(os.environ.get("VOCAB", 1_000_000))
seg = T.Segment(D, V, 10, 42)
dev = T.Device(0)
ix = T.Index.from_segment(dev, seg)
df = np.asarray(seg.terms)[:, 0].astype(np.int64)
kept = np.nonzero(df)[0].astype(np.uint32)
docs, freqs, offs = ix.decode_terms(kept, df[kept])
tf = np.concatenate([[0], np.cumsum(df[kept])]).astype(np.uint64)
f = (freqs & 0xFFFF).astype(np.uint32)
ends = np.cumsum(f)
pos = (np.arange(int(ends[-1]), dtype=np.int64) - np.repeat(ends - f, f) + 1).astype(np.uint16)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    out, terms = dev.encode_google(docs, f, pos, tf)
    best = min(best, time.perf_counter() - t0)
same = out.size == np.asarray(seg.index).size
print(f"encode: {len(docs)} postings, {int(ends[-1])} hits -> {out.size} bytes in {best * 1e3:.1f} ms (two passes + copies both ways): "
      f"{len(docs) / best / 1e6:.0f} M postings/s, {out.size / best / 1e9:.2f} GB/s of index; same size as the segment: {same}")
