"""Query workloads of SURVEY.md §8(d) as postfix programs (include/trinity_hip.h tri_query), shared by bench.py and the tests.

cfg1  `A B`, BM25 top-10 (the reference's own CPU-runnable configuration: 100K documents, 2000 queries)
cfg2  batched `A B`, DocumentsOnly
cfg3  5-term mixed, equal parts `A B (C|D|E)`, `(A|B) (C|D) E`, `A B C D E`, `A|B|C|D|E`; BM25 top-100; lucene_codec
cfg4  phrases `"A B"`, `"A B C"`: half sampled from consecutive slots of a random document, half random terms
cfg5  mixed batch = 50 % cfg2 / 30 % cfg3 / 10 % pure OR / 10 % cfg4 — the cfg3 share scored (BM25 top-100, lucene_codec segment of
      the same corpus), the rest DocumentsOnly on the google_codec segment: two engine batches per step

Terms are Zipf ranks from the segment's own distribution (query seed 1337, distinct within a query)."""
from collections import namedtuple

import numpy as np

from . import engine as E

T, A, O, P = E.OP_TERM, E.OP_AND, E.OP_OR, E.OP_PHRASE

# one engine batch of a workload: programs + execution mode + the codec of the segment it runs on
Part = namedtuple("Part", "name programs flags topk codec")


def _t(x):
    return E.tok(T, int(x))


def and2(rows):
    return [np.array([_t(a), _t(b), E.tok(A, 2)], dtype=np.uint32) for a, b in rows]


def mixed5(rows):
    out = []
    for i, (a, b, c, d, e) in enumerate(rows):
        k = i & 3
        if k == 0:  # A B (C|D|E)
            p = [_t(a), _t(b), _t(c), _t(d), _t(e), E.tok(O, 3), E.tok(A, 3)]
        elif k == 1:  # (A|B) (C|D) E
            p = [_t(a), _t(b), E.tok(O, 2), _t(c), _t(d), E.tok(O, 2), _t(e), E.tok(A, 3)]
        elif k == 2:  # A B C D E
            p = [_t(a), _t(b), _t(c), _t(d), _t(e), E.tok(A, 5)]
        else:  # A|B|C|D|E
            p = [_t(a), _t(b), _t(c), _t(d), _t(e), E.tok(O, 5)]
        out.append(np.array(p, dtype=np.uint32))
    return out


def or5(rows):
    return [np.array([_t(x) for x in r] + [E.tok(O, len(r))], dtype=np.uint32) for r in rows]


def phrases(rows):
    return [np.array([_t(x) for x in r] + [E.tok(P, len(r))], dtype=np.uint32) for r in rows]


def _shuffled(progs, seed):
    """Interleave the query classes so that any strided shard of a part has the same mix."""
    order = np.random.default_rng(seed).permutation(len(progs))
    return [progs[i] for i in order]


def build_parts(name, D, V, slots, corpus_seed, nq, seed=1337):
    """Returns ([Part, …], description)."""
    if name == "cfg1":  # BASELINE.json configs[0]: the reference's CPU-runnable case (S corpus: 100K documents / 10K terms, 2000 queries)
        return [Part("and2-scored", and2(E.gen_queries(V, seed, nq, 2)), E.FLAG_ACCUMULATED_SCORE, 10, E.CODEC_GOOGLE)], "cfg1: 2-term AND, google_codec, AccumulatedScore + BM25, top-10"
    if name == "cfg2":
        return [Part("and2", and2(E.gen_queries(V, seed, nq, 2)), E.FLAG_DOCUMENTS_ONLY, 0, E.CODEC_GOOGLE)], "cfg2: batched 2-term AND, google_codec, DocumentsOnly"
    if name == "cfg3":
        return [Part("mixed5-scored", mixed5(E.gen_queries(V, seed, nq, 5)), E.FLAG_ACCUMULATED_SCORE, 100, E.CODEC_LUCENE)], "cfg3: 5-term mixed AND/OR, lucene_codec (PFOR128), BM25 top-100"
    if name == "cfg4":
        h = nq // 2
        progs = phrases(E.gen_phrase_queries(D, V, slots, corpus_seed, seed, h, 2)) + phrases(E.gen_phrase_queries(D, V, slots, corpus_seed, seed + 1, nq - h, 3))
        return [Part("phrases", progs, E.FLAG_DOCUMENTS_ONLY, 0, E.CODEC_GOOGLE)], 'cfg4: phrases "A B" / "A B C", google_codec, DocumentsOnly'
    if name == "cfg5":
        n2, n3, no = nq // 2, (nq * 3) // 10, nq // 10
        n4 = nq - n2 - n3 - no
        h = n4 // 2
        docs_only = and2(E.gen_queries(V, seed, n2, 2)) + or5(E.gen_queries(V, seed + 2, no, 5))
        docs_only += phrases(E.gen_phrase_queries(D, V, slots, corpus_seed, seed + 3, h, 2)) + phrases(E.gen_phrase_queries(D, V, slots, corpus_seed, seed + 4, n4 - h, 3))
        scored = mixed5(E.gen_queries(V, seed + 1, n3, 5))
        parts = [Part("docsets (50% 2-term AND, 10% 5-way OR, 10% phrases)", _shuffled(docs_only, seed), E.FLAG_DOCUMENTS_ONLY, 0, E.CODEC_GOOGLE),
                 Part("scored (30% 5-term mixed, BM25 top-100)", scored, E.FLAG_ACCUMULATED_SCORE, 100, E.CODEC_LUCENE)]  # fmt: skip
        return parts, "cfg5: mixed batch 50% 2-term AND / 10% 5-way OR / 10% phrases (google_codec, DocumentsOnly) + 30% 5-term mixed (lucene_codec, BM25 top-100)"
    raise ValueError(f"unknown workload {name}")


def build(name, D, V, slots, corpus_seed, nq, seed=1337):
    """Single-part workloads as (programs, flags, topk, codec, description)."""
    parts, desc = build_parts(name, D, V, slots, corpus_seed, nq, seed)
    if len(parts) != 1:
        raise ValueError(f"{name} runs as {len(parts)} engine batches: use build_parts")
    p = parts[0]
    return p.programs, p.flags, p.topk, p.codec, desc
