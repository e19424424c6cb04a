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

# ---- the rest of the write side (round 4): the Lucene-shaped encoder, commit (sort + gather + encode), merge of two halves
def timed(fn, n=2):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        best = min(best, time.perf_counter() - t0)
    return best, r

t, (li, lh, lt) = timed(lambda: dev.encode_lucene(docs, f, pos, tf))
print(f"encode_lucene: {li.size} + {lh.size} bytes (index + hits.data) in {t * 1e3:.1f} ms: {len(docs) / t / 1e6:.0f} M postings/s")
# a session: the same postings in document order (insertion order), the terms named by their index
term_of = np.repeat(kept, df[kept])
order = np.argsort(docs, kind="stable")
hit0 = np.concatenate([[0], ends])[:-1].astype(np.int64)
fo = f[order].astype(np.int64)
take = np.repeat(hit0[order] - np.concatenate([[0], np.cumsum(fo)[:-1]]), fo) + np.arange(int(ends[-1]), dtype=np.int64)
s_pos = pos[take]
s_terms, s_docs, s_freqs = np.ascontiguousarray(term_of[order]), np.ascontiguousarray(docs[order]), np.ascontiguousarray(f[order])
t, (ci, ctids, cterms, cstats) = timed(lambda: dev.commit_google(s_terms, s_docs, s_freqs, s_pos), n=2)
print(f"commit_google: {len(docs)} postings in insertion order -> {ci.size} bytes, {len(ctids)} terms in {t * 1e3:.1f} ms: {len(docs) / t / 1e6:.0f} M postings/s  {cstats}")
