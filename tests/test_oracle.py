"""CPU tests: the oracle (oracle/trinity_oracle.c) against the fixtures produced by the GENUINE reference
(tests/golden/ref_*.json, generator tests/golden/make_golden.py) and, when the prebuilt reference driver is
present (oracle/_ref/ref_driver), against live runs of it."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["tiny", "small", "dense"]


def fnv_bytes(a):
    h = 1469598103934665603
    for x in a.tolist():
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.fixture(scope="module", params=NAMES)
def golden(request):
    with open(os.path.join(GOLDEN, f"ref_{request.param}.json")) as f:
        g = json.load(f)
    c = g["corpus"]
    ix = O.Index.generate(c["D"], c["V"], c["slots"], c["seed"])
    return g, ix


# ---- a1: prefix varint known answers (Switch/switch_compiler_aux.h:23-80) ---------------------------
VARBYTE_KAT = [
    (0, [0x00]),
    (1, [0x01]),
    (127, [0x7F]),
    (128, [0x80, 0x80]),
    (300, [0x81, 0x2C]),
    (16383, [0xBF, 0xFF]),
    (16384, [0xC0, 0x00, 0x40]),
    (0x12345, [0xC1, 0x45, 0x23]),
    (2097151, [0xDF, 0xFF, 0xFF]),
    (2097152, [0xE0, 0x20, 0x00, 0x00]),
    (0x0ABCDEF1, [0xEA, 0xBC, 0xDE, 0xF1]),
    (268435455, [0xEF, 0xFF, 0xFF, 0xFF]),
    (268435456, [0xF0, 0x00, 0x00, 0x00, 0x10]),
    (0xFFFFFFFF, [0xF0, 0xFF, 0xFF, 0xFF, 0xFF]),
]


def test_varbyte_kat():
    import ctypes as C

    L = O.lib()
    for v, enc in VARBYTE_KAT:
        buf = (C.c_uint8 * 8)()
        n = L.to_varbyte_put32(buf, v)
        assert list(buf[:n]) == enc, (v, list(buf[:n]))
        out = C.c_uint32()
        m = L.to_varbyte_get32(buf, C.byref(out))
        assert (m, out.value) == (n, v)


def test_varbyte_roundtrip_random():
    import ctypes as C

    L = O.lib()
    rng = np.random.default_rng(1)
    vals = np.concatenate([rng.integers(0, 1 << b, 200, dtype=np.uint64) for b in (7, 8, 14, 15, 21, 22, 28, 29, 32)])
    buf = (C.c_uint8 * 8)()
    out = C.c_uint32()
    for v in vals.tolist():
        n = L.to_varbyte_put32(buf, v)
        assert L.to_varbyte_get32(buf, C.byref(out)) == n and out.value == v


def test_index_bytes_match_reference(golden):
    g, ix = golden
    r = g["results"][0]
    assert r["cmd"] == "index"
    assert ix.c.len == r["len"]
    assert str(fnv_bytes(ix.bytes())) == r["fnv"]
    assert str(O.fnv1a_docs(ix.terms().reshape(-1))) == r["terms_fnv"]
    assert int(ix.c.sumTermsDocs) == r["postings"] and int(ix.c.totalTerms) == r["totalTerms"]


def test_chunk_walker_accounts_for_every_byte(golden):
    _, ix = golden
    tot = 0
    for t in range(ix.c.nterms):
        s = ix.chunk_stats(t)
        tot += s["hdr"] + s["docfreq"] + s["hits"] + s["skip"]
        assert s["postings"] == ix.c.terms[t].documents
    assert tot == ix.c.len


def test_decode_next_matches_reference(golden):
    g, ix = golden
    n = 0
    for r in g["results"]:
        if r["cmd"] != "decode":
            continue
        d, f = ix.decode_term(r["term"])
        assert len(d) == r["n"]
        assert str(O.fnv1a_docs(d)) == r["docs_fnv"] and str(O.fnv1a_docs(f)) == r["freqs_fnv"]
        k = min(8, len(d))
        assert d[:k].tolist() == r["first"] and d[len(d) - k :].tolist() == r["last"]
        n += 1
    assert n >= 5


def _advance_trace(ix, D, term, seed, steps):
    import ctypes as C

    L = O.lib()
    it = O.PLI(ix, term)
    st = C.c_uint64(seed)
    vals = []
    done = 0
    while done < steps and it.current() != O.DOCIDS_END:
        r = L.to_splitmix64(C.byref(st))
        cur = it.current()
        op = r & 7
        j = r >> 8
        if op in (0, 1):
            i = it.next()
        elif op == 2:
            i = it.advance(cur) if cur else it.next()
        elif op == 3:
            i = it.advance(cur + 1 + j % 3)
        elif op == 4:
            i = it.advance(cur + 1 + j % 64)
        elif op == 5:
            i = it.advance(cur + 1 + j % 2048)
        elif op == 6:
            i = it.advance(cur + 1 + j % (D // 16 + 1))
        else:
            i = it.advance(cur - j % 3) if cur > 3 else it.next()
        vals += [i, 0 if i == O.DOCIDS_END else it.freq()]
        done += 1
    return done, O.fnv1a_u32_stream(np.array(vals, dtype=np.uint32))


def test_advance_traces_match_reference(golden):
    g, ix = golden
    D = g["corpus"]["D"]
    n = 0
    for r in g["results"]:
        if r["cmd"] != "advance":
            continue
        done, h = _advance_trace(ix, D, r["term"], int(r["seed"]), r["steps"])
        assert done == r["done"] and str(h) == r["trace_fnv"], r
        n += 1
    assert n >= 6


def test_positions_match_reference(golden):
    g, ix = golden
    for r in g["results"]:
        if r["cmd"] != "positions":
            continue
        it = O.PLI(ix, r["term"])
        vals = []
        i = 0
        cnt = 0
        while True:
            d = it.next()
            if d == O.DOCIDS_END:
                break
            if i % r["nth"] == 0:
                f = it.freq()
                pos = it.positions()
                assert len(pos) == f
                vals += [d] + pos
                cnt += 1
            i += 1
        assert cnt == r["docs"] and str(O.fnv1a_u32_stream(np.array(vals, dtype=np.uint32))) == r["fnv"]


def test_queries_match_reference(golden):
    g, ix = golden
    n = 0
    for r in g["results"]:
        if r["cmd"] not in ("query", "queryfull") or r["flags"] == 0:
            continue
        ix.set_similarity({"bm25": O.SIM_BM25, "tfidf": O.SIM_TFIDF, "trivial": O.SIM_TRIVIAL}[r.get("sim", "bm25")])
        docs, scores = ix.exec(O.parse_query(r["q"]), r["flags"])
        assert len(docs) == r["n"], r["q"]
        assert str(O.fnv1a_docs(docs)) == r["fnv"], r["q"]  # bit-exact docID set, ascending
        k = min(16, len(docs))
        assert docs[:k].tolist() == r["first"] and docs[len(docs) - k :].tolist() == r["last"]
        if r["flags"] & 2:
            ref_sum = r["score_sum"]
            assert abs(scores.sum() - ref_sum) <= 1e-5 * max(1.0, abs(ref_sum))
            if "top" in r:
                td, ts = ix.topk(docs, scores, len(r["top"]))
                assert td.tolist() == [x[0] for x in r["top"]]
                np.testing.assert_allclose(ts, [x[1] for x in r["top"]], rtol=1e-5)
            if "scores" in r:
                assert docs.tolist() == r["docs"]
                np.testing.assert_allclose(scores, r["scores"], rtol=1e-5, atol=0)
        elif "docs" in r:
            assert docs.tolist() == r["docs"]
        n += 1
    ix.set_similarity(O.SIM_BM25)
    assert n > 200


def test_matchsome_matches_reference(golden):
    """matchsome -> DocsSetIterators::DisjunctionSome (docset_iterators.cpp:679-811, the two-heap min-should-match iterator):
    docID sets, BM25 sums / top-10 and the default mode's matched terms, thresholds 2..n, nested under AND and over
    sub-expressions.  Oracle only: the GPU planner does not lower it yet."""
    g, ix = golden
    n = 0
    for r in g["results"]:
        if r["cmd"] != "querysome":
            continue
        p = O.parse_query(r["q"], some_min=r["min"])
        if r["flags"] == 0:
            docs, flat, tt, ht = ix.exec_rich(p)
            assert len(docs) == r["n"] and str(O.fnv1a_docs(docs)) == r["fnv"], (r["q"], r["min"])
            assert (tt, ht) == (r["terms_total"], r["hits_total"]) and str(O.fnv1a_u32_stream(flat)) == r["rich_fnv"], (r["q"], r["min"])
        else:
            docs, scores = ix.exec(p, r["flags"])
            assert len(docs) == r["n"] and str(O.fnv1a_docs(docs)) == r["fnv"], (r["q"], r["min"])
            if r["flags"] & 2:
                assert abs(scores.sum() - r["score_sum"]) <= 1e-5 * max(1.0, abs(r["score_sum"]))
                if "top" in r:
                    td, ts = ix.topk(docs, scores, len(r["top"]))
                    assert td.tolist() == [x[0] for x in r["top"]]
                    np.testing.assert_allclose(ts, [x[1] for x in r["top"]], rtol=1e-5)
        n += 1
    assert n >= 150


def test_rich_mode_matches_reference(golden):
    """exec_query's default mode (flags 0): per match the query terms that matched and their hits — the canonical stream's
    FNV, the totals and the first documents in full, against the genuine reference."""
    g, ix = golden
    n = 0
    for r in g["results"]:
        if r["cmd"] != "query" or r["flags"] != 0:
            continue
        docs, flat, tt, ht = ix.exec_rich(O.parse_query(r["q"]))
        assert len(docs) == r["n"] and str(O.fnv1a_docs(docs)) == r["fnv"], r["q"]
        assert tt == r["terms_total"] and ht == r["hits_total"], r["q"]
        assert str(O.fnv1a_u32_stream(flat)) == r["rich_fnv"], r["q"]
        at = 0
        for want in r["rich_docs"]:
            assert flat[at : at + len(want)].tolist() == want, (r["q"], want)
            at += len(want)
        n += 1
    assert n >= 100


def test_bm25_formula_known_answers():
    L = O.lib()
    # similarity.h:179-181 float-precision idf; :228-235 score
    import math

    N, df = 20000, 17135
    f32 = np.float32
    idf = float(np.log(f32(1) + (f32(N - df) + f32(0.5)) / (f32(df) + f32(0.5)), dtype=np.float32))
    assert L.to_bm25_idf(df, N) == pytest.approx(idf, rel=1e-7)
    for fr in (1, 2, 3, 10, 65535):
        want = f32(idf * float(f32(fr)) / float(f32(fr) + f32(1.2)))
        assert L.to_bm25_score(idf, fr) == pytest.approx(float(want), rel=1e-6)
    assert math.isfinite(L.to_bm25_idf(0, 1))


@pytest.mark.skipif(not os.path.exists(O.REF_DRIVER), reason="prebuilt reference driver not present")
def test_live_reference_random_queries():
    """Live cross-check at a size the fixtures do not hold: 50K docs, seeded random 2..5-term shapes."""
    D, V, S, seed = 50000, 3000, 10, 99
    ix = O.Index.generate(D, V, S, seed)
    qs = O.gen_queries(V, 4242, 40, 5).tolist()
    tpls = ["t{a} t{b}", "t{a} OR t{b} OR t{c}", "t{a} t{b} (t{c} OR t{d} OR t{e})", '"t{a} t{b}"', "(t{a} OR t{b}) (t{c} OR t{d}) t{e}"]
    cmds, meta = ["index"], []
    for i, row in enumerate(qs):
        a, b, c, d, e = row
        text = tpls[i % len(tpls)].format(a=a, b=b, c=c, d=d, e=e)
        for fl in (1, 2):
            cmds.append(f"query {fl} {10 if fl == 2 else 0} {text}")
    res = O.run_ref_driver(D, V, S, seed, cmds)
    assert str(fnv_bytes(ix.bytes())) == res[0]["fnv"]
    for r in res[1:]:
        docs, scores = ix.exec(O.parse_query(r["q"]), r["flags"])
        assert len(docs) == r["n"] and str(O.fnv1a_docs(docs)) == r["fnv"], r["q"]
        if r["flags"] & 2:
            assert abs(scores.sum() - r["score_sum"]) <= 1e-5 * max(1.0, abs(r["score_sum"]))


# ------------------------------------------------------------------ Lucene-shaped codec (PFOR payload parity unpinned)
def test_ints_group_roundtrip_and_known_answers():
    """ints() groups (lucene_codec.cpp:26-100): all-equal rule is the reference's; the packed payload is this repo's
    PFOR128 (include/pfor128.md)."""
    L = O.lib()
    buf = np.zeros(2048, np.uint8)
    out = np.zeros(128, np.uint32)
    v = np.full(128, 300, np.uint32)  # all equal: u8 0 + prefix-varint(300) = 00 81 2C
    assert L.to_ints_encode(v.ctypes.data, buf.ctypes.data) == 3 and buf[:3].tolist() == [0x00, 0x81, 0x2C]
    v = (np.arange(128) & 3).astype(np.uint32)  # width 2, no exceptions: L = 1 + 4*2 words
    n = L.to_ints_encode(v.ctypes.data, buf.ctypes.data)
    assert n == 1 + 4 * 9 and buf[0] == 9 and buf[1:5].tolist() == [2, 0, 0, 0] and buf[5] == 0b11100100
    rng = np.random.default_rng(5)
    for trial in range(1500):
        kind = trial % 5
        if kind == 0:
            v = rng.integers(0, 4, 128)
        elif kind == 1:
            v = rng.geometric(0.2, 128)
        elif kind == 2:
            v = np.where(rng.random(128) < 0.08, rng.integers(0, 1 << 31, 128), rng.integers(0, 16, 128))
        elif kind == 3:
            v = rng.integers(0, 1 << 32, 128)
        else:
            v = rng.integers(0, 1 << rng.integers(1, 32), 128)
        v = v.astype(np.uint32)
        n = L.to_ints_encode(v.ctypes.data, buf.ctypes.data)
        assert L.to_ints_decode(buf.ctypes.data, out.ctypes.data) == n and np.array_equal(v, out)
        assert buf[0] == 0 or n == 1 + 4 * int(buf[0])


@pytest.fixture(scope="module", params=NAMES)
def both_codecs(request):
    with open(os.path.join(GOLDEN, f"ref_{request.param}.json")) as f:
        g = json.load(f)
    c = g["corpus"]
    return g, O.Index.generate(c["D"], c["V"], c["slots"], c["seed"]), O.Index.generate(c["D"], c["V"], c["slots"], c["seed"], codec="lucene")


def test_lucene_codec_results_equal_reference_fixtures(both_codecs):
    """The genuine reference could only be built with its Google codec here; a Lucene-coded segment of the same corpus
    must give the same docID sets and scores (the reference's codecs are interchangeable behind Codecs::Decoder)."""
    g, _, lx = both_codecs
    n = 0
    for r in g["results"]:
        if r["cmd"] == "decode":
            d, f = lx.decode_term(r["term"])
            assert str(O.fnv1a_docs(d)) == r["docs_fnv"] and str(O.fnv1a_docs(f)) == r["freqs_fnv"]
        elif r["cmd"] in ("query", "queryfull") and r["flags"] == 0:
            docs, flat, tt, ht = lx.exec_rich(O.parse_query(r["q"]))
            assert len(docs) == r["n"] and str(O.fnv1a_u32_stream(flat)) == r["rich_fnv"], r["q"]
        elif r["cmd"] in ("query", "queryfull"):
            lx.set_similarity({"bm25": O.SIM_BM25, "tfidf": O.SIM_TFIDF, "trivial": O.SIM_TRIVIAL}[r.get("sim", "bm25")])
            docs, scores = lx.exec(O.parse_query(r["q"]), r["flags"])
            assert len(docs) == r["n"] and str(O.fnv1a_docs(docs)) == r["fnv"], r["q"]
            if r["flags"] & 2:
                assert abs(scores.sum() - r["score_sum"]) <= 1e-5 * max(1.0, abs(r["score_sum"]))
            n += 1
        elif r["cmd"] == "positions":
            it = O.PLI(lx, r["term"])
            vals, i = [], 0
            while True:
                d = it.next()
                if d == O.DOCIDS_END:
                    break
                if i % r["nth"] == 0:
                    vals += [d] + it.positions()
                i += 1
            assert str(O.fnv1a_u32_stream(np.array(vals, dtype=np.uint32))) == r["fnv"]
        elif r["cmd"] == "advance":
            D = g["corpus"]["D"]
            done, h = _advance_trace(lx, D, r["term"], int(r["seed"]), r["steps"])
            assert done == r["done"] and str(h) == r["trace_fnv"], r  # same (doc, freq) after every next()/advance()
    assert n > 200


# ---- the EDGE segment: reference-PRODUCED bytes (payload-bearing hits, position 0, freq 0, a wrapped frequency, MaxPosition) ----
def load_edge():
    import base64

    with open(os.path.join(GOLDEN, "ref_edge.json")) as f:
        g = json.load(f)
    index = np.frombuffer(base64.b64decode(g["index_b64"]), dtype=np.uint8)
    terms = np.array(g["terms"], dtype=np.uint32)
    return g, index, terms


@pytest.fixture(scope="module")
def edge():
    g, index, terms = load_edge()
    return g, O.Index.wrap(index, terms, g["docsCnt"], g["postings"], g["sumTermHits"])


def test_edge_segment_bytes_are_the_reference_s(edge):
    g, ix = edge
    r = g["results"][0]
    assert r["cmd"] == "index" and r["len"] == len(ix.bytes()) and str(fnv_bytes(ix.bytes())) == r["fnv"]


def test_edge_decode_and_hits_match_reference(edge):
    """next() to exhaustion (freq as the iterator exposes it: tokenpos_t — the 70000-hit document reads 4464, a document whose
    only hit sits at position 0 without payload reads 0) and materialize_hits positions across payloads of changing length."""
    g, ix = edge
    n = 0
    for r in g["results"]:
        if r["cmd"] == "decode":
            d, f = ix.decode_term(r["term"])
            assert len(d) == r["n"] and str(O.fnv1a_docs(d)) == r["docs_fnv"], r["term"]
            assert str(O.fnv1a_docs(f & 0xFFFF)) == r["freqs_fnv"], r["term"]
            n += 1
        elif r["cmd"] == "hits":
            it = O.PLI(ix, r["term"])
            vals, docs = [], 0
            while True:
                i = it.next()
                if i == O.DOCIDS_END:
                    break
                f = it.freq()
                pos = it.positions()
                assert len(pos) == f, (r["term"], i, f, len(pos))
                vals += [f, i] + pos
                docs += 1
            # ref_driver: hpos = fnv_u32(id, fnv_u32(f, hpos)) then the positions
            assert docs == r["docs"] and str(O.fnv1a_u32s(vals)) == r["pos_fnv"], r["term"]
            n += 1
    assert n >= 14
    d, f = ix.decode_term(2)
    assert int(f[d.tolist().index(100)]) & 0xFFFF == 70000 & 0xFFFF  # the wrapped frequency
    d, f = ix.decode_term(1)
    assert (f == 0).sum() > 100  # documents of frequency 0


def test_edge_advance_traces_match_reference(edge):
    g, ix = edge
    n = 0
    for r in g["results"]:
        if r["cmd"] == "advance":
            done, h = _advance_trace(ix, g["docsCnt"], r["term"], int(r["seed"]), r["steps"])
            assert done == r["done"] and str(h) == r["trace_fnv"], r
            n += 1
    assert n == 8


def test_edge_queries_match_reference(edge):
    g, ix = edge
    n = 0
    for r in g["results"]:
        if r["cmd"] == "queryfull":
            docs, scores = ix.exec(O.parse_query(r["q"]), r["flags"])
            assert docs.tolist() == r["docs"], (r["q"], r["flags"])
            if r["flags"] & 2:
                np.testing.assert_allclose(scores, r.get("scores", []), rtol=1e-5, atol=0)  # (an empty result prints no scores)
            n += 1
        elif r["cmd"] == "query" and r["flags"] == 0:
            docs, flat, tt, ht = ix.exec_rich(O.parse_query(r["q"]))
            assert len(docs) == r["n"] and str(O.fnv1a_docs(docs)) == r["fnv"], r["q"]
            assert (tt, ht) == (r["terms_total"], r["hits_total"]) and str(O.fnv1a_u32_stream(flat)) == r["rich_fnv"], r["q"]
            n += 1
    assert n >= 40


def test_compiled_exec_trees_lower_to_the_reference_s_answers():
    """tests/golden/ref_trees.json holds what the reference's compile_query makes of 165 queries (the exec_node trees build_iterator
    consumes) and the reference's answers.  Lowered node by node to postfix programs (oracle_lib.program_from_exec_tree), the oracle
    must reproduce those answers: the lowering — the same one the GPU tests feed the C-ABI with — is pinned on the CPU."""
    g = json.load(open(os.path.join(GOLDEN, "ref_trees.json")))
    c = g["corpus"]
    ix = O.Index.generate(c["D"], c["V"], c["slots"], c["seed"])
    shapes = set()
    for r in g["results"]:
        prog = np.array(O.program_from_exec_tree(r["tree"]), dtype=np.uint32)
        shapes.add(json.dumps(r["tree"]).count('"op"'))
        docs, _ = ix.exec(prog, O.FLAG_DOCUMENTS_ONLY)
        assert len(docs) == r["n"] and str(O.fnv1a_docs(docs)) == r["fnv"], r["q"]
        docs, scores = ix.exec(prog, O.FLAG_ACCUM_SCORE)
        assert len(docs) == r["n"], r["q"]
        assert abs(float(np.sum(scores)) - r["score_sum"]) <= 1e-6 * max(1.0, r["score_sum"]), r["q"]
        td, ts = ix.topk(docs, scores, 10)
        assert td.tolist() == [x[0] for x in r["top"]], r["q"]
        np.testing.assert_allclose(ts, [x[1] for x in r["top"]], rtol=1e-6)
    assert len(g["results"]) >= 150 and len(shapes) >= 4


def test_random_trees_equal_the_reference():
    """tests/golden/ref_random.json: 384 random trees of AND / OR / NOT / <optional> / matchsome (depth <= 3, <= 8 distinct terms) as the
    reference compiled them, with its answers in all three modes.  The oracle's iterators, fed the lowered trees, reproduce them — docID
    sets, score sums and top-10, and the default mode's matched terms and hits."""
    g = json.load(open(os.path.join(GOLDEN, "ref_random.json")))
    c = g["corpus"]
    ix = O.Index.generate(c["D"], c["V"], c["slots"], c["seed"])
    ops = set()
    for r in g["results"]:
        prog = np.array(O.program_from_exec_tree(r["tree"]), dtype=np.uint32)
        ops |= {int(t) >> 28 for t in prog}
        docs, _ = ix.exec(prog, O.FLAG_DOCUMENTS_ONLY)
        assert len(docs) == r["n"] and str(O.fnv1a_docs(docs)) == r["fnv"], r["q"]
        docs, scores = ix.exec(prog, O.FLAG_ACCUM_SCORE)
        assert len(docs) == r["n"] and abs(float(np.sum(scores)) - r["score_sum"]) <= 1e-6 * max(1.0, r["score_sum"]), r["q"]
        td, ts = ix.topk(docs, scores, 10)
        assert td.tolist() == [x[0] for x in r["top"]], r["q"]
        wdocs, wflat, tt, ht = ix.exec_rich(prog)
        assert len(wdocs) == r["n"] and tt == r["terms_total"] and ht == r["hits_total"], r["q"]
        assert str(O.fnv1a_u32_stream(wflat)) == r["rich_fnv"], r["q"]
    assert len(g["results"]) >= 350 and ops >= {O.OP_TERM, O.OP_AND, O.OP_OR, O.OP_NOT, O.OP_OPT, O.OP_SOME}


def test_phrases_inside_trees_equal_the_reference():
    """tests/golden/ref_phrase_trees.json: a multi-word phrase under an OR, inside a matchsome, under a NOT / an <optional> — 240 trees as the
    reference compiled them (tiny and dense corpus), with its answers in AccumulatedScore and default mode (DocumentsOnly crashes the reference
    on these).  The oracle's iterators reproduce them: Phrase is an iterator like any other (docset_iterators.cpp:66-224).
    (The GPU planner still leaves these shapes out, per query — tests/test_gpu_parity.py::test_shapes_still_refused; this pins what it will
    have to return.)"""
    g = json.load(open(os.path.join(GOLDEN, "ref_phrase_trees.json")))
    ixs = {name: O.Index.generate(c["D"], c["V"], c["slots"], c["seed"]) for name, c in g["corpora"].items()}
    ops, nonempty, hashed = set(), 0, 0
    for r in g["results"]:
        ix = ixs[r["corpus"]]
        prog = np.array(O.program_from_exec_tree(r["tree"]), dtype=np.uint32)
        ops |= {int(t) >> 28 for t in prog}
        docs, scores = ix.exec(prog, O.FLAG_ACCUM_SCORE)
        assert len(docs) == r["n"] and abs(float(np.sum(scores)) - r["score_sum"]) <= 1e-6 * max(1.0, r["score_sum"]), r["q"]
        td, ts = ix.topk(docs, scores, 10)
        assert td.tolist() == [x[0] for x in r["top"]], r["q"]
        np.testing.assert_allclose(ts, [x[1] for x in r["top"]], rtol=1e-5, atol=0)
        wdocs, wflat, tt, ht = ix.exec_rich(prog)
        assert len(wdocs) == r["rich_n"] and tt == r["terms_total"] and ht == r["hits_total"], r["q"]
        if r["rich_fnv"] is not None:  # (None: a shape whose default-mode positions the reference itself gets wrong — make_golden.py says which, by rule)
            assert str(O.fnv1a_u32_stream(wflat)) == r["rich_fnv"], r["q"]
            hashed += 1
        nonempty += r["n"] > 0
    assert hashed >= 180
    assert len(g["results"]) == 240 and nonempty >= 200 and ops >= {O.OP_TERM, O.OP_AND, O.OP_OR, O.OP_NOT, O.OP_OPT, O.OP_SOME, O.OP_PHRASE}


def test_edge_hit_payloads_match_reference(edge):
    """`hits <term>` records carry the reference's hash over every document's (freq, id) and every hit's (pos, payloadLen, the eight bytes
    of term_hit::payload) — payload lengths that change from hit to hit, the stale high bytes a shorter payload leaves in the word."""
    g, ix = edge
    n = 0
    for r in g["results"]:
        if r["cmd"] != "hits":
            continue
        it = O.PLI(ix, r["term"])
        h = 1469598103934665603
        seen_payload = False
        while True:
            i = it.next()
            if i == O.DOCIDS_END:
                break
            f = it.freq()
            pos, ln, pl = it.hits()
            assert len(pos) == f
            h = O.fnv1a_u32s([f, i], h)
            for p_, l_, w_ in zip(pos, ln, pl):
                h = O.fnv1a_u32s([p_, l_, w_ & 0xFFFFFFFF, w_ >> 32], h)
                seen_payload |= l_ != 0
        assert str(h) == r["fnv"], r["term"]
        n += seen_payload
    assert n >= 1


def test_documents_ruled_out_before_consider_equal_the_reference():
    """tests/golden/ref_masked.json: the reference run with a rule-backed IndexDocumentsFilter (`filter <seed> <permille>`, matches.h:198-201)
    — the hook exec_query tests in the same condition as masked_documents_registry::test, right before consider(), in every execution mode
    (exec.cpp:1095-1150 ...).  The oracle with the same documents installed as the segment's MASKED set gives the reference's docID sets
    (DocumentsOnly), score sums and top-10 (AccumulatedScore: a dropped document leaves the ranking), matched terms and hits (default
    mode) — 1524 records over two corpora and three drop rates (5 %, 30 %, 90 %): the masking semantics of SURVEY §8(f2) pinned to
    reference code."""
    g = json.load(open(os.path.join(GOLDEN, "ref_masked.json")))
    ixs = {name: O.Index.generate(c["D"], c["V"], c["slots"], c["seed"]) for name, c in g["corpora"].items()}
    n, by_mode, dropped_any = 0, {0: 0, 1: 0, 2: 0}, 0
    cur = None
    for r in g["results"]:
        ix, c = ixs[r["corpus"]], g["corpora"][r["corpus"]]
        key = (r["corpus"], tuple(r["filter"]))
        if key != cur:
            with np.errstate(over="ignore"):
                ix.set_masked(O.masked_docs(c["D"], *r["filter"]))
            cur = key
        prog = O.parse_query(r["q"], some_min=r["min"] or 1)
        if r["flags"] == 0:
            docs, flat, tt, ht = ix.exec_rich(prog)
            assert len(docs) == r["n"] and str(O.fnv1a_docs(docs)) == r["fnv"], r["q"]
            assert tt == r["terms_total"] and ht == r["hits_total"] and str(O.fnv1a_u32_stream(flat)) == r["rich_fnv"], r["q"]
        else:
            docs, scores = ix.exec(prog, r["flags"])
            assert len(docs) == r["n"] and str(O.fnv1a_docs(docs)) == r["fnv"], (r["q"], r["filter"])
            if r["flags"] & 2:
                assert abs(scores.sum() - r["score_sum"]) <= 1e-5 * max(1.0, abs(r["score_sum"]))
                td, ts = ix.topk(docs, scores, len(r.get("top", [])))
                assert td.tolist() == [x[0] for x in r.get("top", [])], r["q"]
                np.testing.assert_allclose(ts, [x[1] for x in r.get("top", [])], rtol=1e-5)
        by_mode[r["flags"]] += 1
        n += 1
    for ix in ixs.values():
        ix.set_masked(np.zeros(0, np.uint32))
    assert n == len(g["results"]) >= 1500 and min(by_mode.values()) >= 400
