"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against the CPU
oracle on the same seeded inputs, against the fixtures produced by the genuine reference, and — at sizes the
oracle cannot cover quickly — through size-independent properties."""
import contextlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def T():
    import trinity_amd

    trinity_amd.build_all()
    return trinity_amd


@pytest.fixture(scope="module")
def dev(T):
    from conftest import apply_test_options

    d = apply_test_options(T.Device(0))
    yield d
    d.close()


@contextlib.contextmanager
def options(dev, **kw):
    """Planner options of the device handle (tri_dev_set_option) for the batches created inside the block."""
    old = {k: dev.get_option(k) for k in kw}
    for k, v in kw.items():
        dev.set_option(k, v)
    try:
        yield
    finally:
        for k, v in old.items():
            dev.set_option(k, v)


ALL_PLANES = 1 << 30  # plane_div: every term of a batch gets a plane
# term planes (k_planes.hpp): the planner's choice (head terms of at least docs / 64 documents), a plane for every term, no term
# planes at all (k_planes decodes every slot into LDS planes; queries of more than six slots stay with k_fused), everything off
PLANE_SETS = ({}, {"plane_div": ALL_PLANES}, {"plane_div": 0}, {"planes": 0}, {"plane_div": ALL_PLANES, "probe_max_blocks": 1 << 20})  # (the last: every single-lead conjunction through k_probe)


class World:
    def __init__(self, T, dev, D, V, slots, seed, codec=1):
        self.T, self.D, self.V, self.dev = T, D, V, dev
        self.seg = T.Segment(D, V, slots, seed, codec=codec)
        if codec == 1:
            self.ora = O.Index.wrap(self.seg.index, self.seg.terms, self.seg.docs_cnt, self.seg.sum_terms_docs, self.seg.sum_term_hits)
        else:
            # LUCENE-shaped segment on the GPU; the checker is the oracle's own Lucene-coded index of the same corpus
            # (byte-identical to the product builder's, tests/test_abi.py), whose results equal the Google-coded ones
            self.ora = O.Index.generate(D, V, slots, seed, codec="lucene")
        self.ix = T.Index.from_segment(dev, self.seg)

    def df(self, t):
        return int(self.seg.terms[t, 0]) if t < self.V else 0


@pytest.fixture(scope="module")
def small(T, dev):
    return World(T, dev, 20000, 2000, 10, 42)


@pytest.fixture(scope="module")
def medium(T, dev):
    return World(T, dev, 300000, 30000, 10, 42)


@pytest.fixture(scope="module")
def dense(T, dev):
    return World(T, dev, 20000, 500, 12, 7)


@pytest.fixture(scope="module")
def longdocs(T, dev):
    """Documents of 90 token slots: a head term's hits run past position 64 and past seven a document — k_phrase's three ways to a candidate's
    hits (the entry holds them / a locator / the byte stream) and both of its checks (the 64-bit position words, the walk) in one corpus."""
    return World(T, dev, 6000, 400, 90, 11)


def gpu_lowers(q, rich=False):
    """Every fixture shape is lowered in DocumentsOnly and AccumulatedScore top-K mode (NOT of an AND — `a NOT (b c)` — runs off a
    truth table); the default (matched terms) mode still answers TRI_ERR_UNSUPPORTED to that one."""
    if not rich or "NOT (" not in q:
        return True
    return " OR " in q.split("NOT (", 1)[1]


def run_docs_only(w, programs):
    b = w.T.Batch(w.ix, programs, w.T.FLAG_DOCUMENTS_ONLY)
    b.run()
    b.sync()
    counts = b.counts()
    sets = [b.docset(i, int(counts[i])) for i in range(len(programs))]
    hashes = b.docset_hashes()
    info = b.info()
    b.close()
    return sets, hashes, info


def and_prog(T, terms):
    return np.array([T.tok(T.OP_TERM, t) for t in terms] + [T.tok(T.OP_AND, len(terms))], dtype=np.uint32)


# ------------------------------------------------------------------------------------------ decode (K1)
@pytest.mark.parametrize("world", ["small", "dense", "medium"])
def test_decode_terms_bit_exact(request, world):
    w = request.getfixturevalue(world)
    terms = [0, 1, 2, 3, 5, 17, w.V // 3, w.V // 2, w.V - 1]
    terms = [t for t in terms if w.df(t)]
    docs, freqs, offs = w.ix.decode_terms(terms, [w.df(t) for t in terms])
    for i, t in enumerate(terms):
        d, f = w.ora.decode_term(t)
        assert np.array_equal(docs[offs[i] : offs[i + 1]], d), t
        assert np.array_equal(freqs[offs[i] : offs[i + 1]], f), t


def test_directory_accounts_for_every_chunk_byte(small):
    info = small.ix.info()
    skip = sum(small.ora.chunk_stats(t)["skip"] for t in range(small.V))
    assert info["doc_bytes"] + info["hit_bytes"] + skip == info["index_bytes"] == small.seg.index.size
    assert info["postings"] == small.seg.sum_terms_docs
    db = small.ix.term_docbytes(np.arange(8))
    for t in range(8):
        s = small.ora.chunk_stats(t)
        assert db[t] == s["hdr"] + s["docfreq"]  # SURVEY §8(d) docbytes(t)


# ------------------------------------------------------------------------------------------ AND (K3)
@pytest.mark.parametrize("world,nq", [("small", 400), ("dense", 300), ("medium", 250)])
def test_and2_matches_oracle(request, world, nq):
    w = request.getfixturevalue(world)
    T = w.T
    qs = T.gen_queries(w.V, 1337, nq, 2).tolist() + [[0, 1], [1, 0], [0, 2], [0, w.V - 1], [1, w.V // 2], [3, 4]]
    progs = [and_prog(T, q) for q in qs]
    sets, hashes, info = run_docs_only(w, progs)
    tot = 0
    for q, got, h in zip(qs, sets, hashes):
        want, _ = w.ora.exec(and_prog(T, q), O.FLAG_DOCUMENTS_ONLY)
        assert np.array_equal(got, want), (q, len(got), len(want))
        assert int(h) == O.fnv1a_docs(want)
        tot += len(want)
    assert info["matches"] == tot


@pytest.fixture(scope="module")
def large(T, dev):
    return World(T, dev, 2_000_000, 200_000, 10, 42)


def test_and_dense_windows_match_oracle(large):
    """Head x head conjunctions on a 2M-document segment run as TASK_DENSE (bitmap windows, many tasks/query)."""
    w, T = large, large.T
    qs = [[0, 1], [1, 0], [0, 2], [1, 2], [0, 7], [3, 5], [0, 1, 2], [2, 4, 1, 0], [0, 25], [1, 40]]
    progs = [and_prog(T, q) for q in qs]
    sets, hashes, info = run_docs_only(w, progs)
    for q, got, h in zip(qs, sets, hashes):
        want, _ = w.ora.exec(and_prog(T, q), O.FLAG_DOCUMENTS_ONLY)
        assert np.array_equal(got, want), (q, len(got), len(want))
        assert int(h) == O.fnv1a_docs(want)


@pytest.mark.parametrize("world,nq", [("small", 300), ("dense", 200)])
def test_and_forced_dense_path_matches_oracle(request, world, nq):
    """dense_min_postings = 0 forces every eligible query through the bitmap-window path: sparse windows, windows
    with no blocks, lists ending mid-window, k-way conjunctions."""
    w = request.getfixturevalue(world)
    T = w.T
    qs = T.gen_queries(w.V, 7, nq, 2).tolist() + T.gen_queries(w.V, 8, 60, 3).tolist() + [[0, 1], [0, 1, 2, 3], [w.V - 1, 0]]
    wants = [w.ora.exec(and_prog(T, q), O.FLAG_DOCUMENTS_ONLY)[0] for q in qs]
    for ps in PLANE_SETS:  # the bitmap windows with the planner's term planes, with a plane for every term, without any
        with options(w.dev, dense_min_postings=0, **ps):
            sets, _, info = run_docs_only(w, [and_prog(T, q) for q in qs])
            planes_on = w.dev.get_option("planes") != 0 and w.dev.get_option("plane_div") != 0  # (TRINITY_TEST_OPTIONS may have set them for the whole run)
        assert (info["plane_terms"] > 0) == planes_on, ps
        for q, got, want in zip(qs, sets, wants):
            assert np.array_equal(got, want), (ps, q, len(got), len(want))


@pytest.mark.parametrize("world", ["small", "dense", "small_l"])
def test_candidate_tiles_probe_term_planes(request, world):
    """k_and with the probed lists read from term planes (one bit per candidate) instead of bracket + block decode: 2- and 3-term
    conjunctions, a NOT, an OR group behind a sparse lead — with a plane for every term and with the planner's choice; equal to the
    oracle and to the run without planes."""
    w = request.getfixturevalue(world)
    T = w.T
    texts = [f"t{a} t{b}" for a, b in T.gen_queries(w.V, 21, 200, 2).tolist()] + [f"t{a} t{b} t{c}" for a, b, c in T.gen_queries(w.V, 22, 80, 3).tolist()]
    texts += [f"t{a} t{b} NOT t{c}" for a, b, c in T.gen_queries(w.V, 23, 40, 3).tolist()] + [f"t{a} (t{b} OR t{c} OR t0)" for a, b, c in T.gen_queries(w.V, 24, 40, 3).tolist()]
    progs = [O.parse_query(t) for t in texts]
    wants = [w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)[0] for p in progs]
    for ps in ({"planes": 1, "plane_div": ALL_PLANES}, {"planes": 1}, {"planes": 0}, {"planes": 1, "plane_div": ALL_PLANES, "probe_max_blocks": 1 << 20},
               {"planes": 1, "plane_div": ALL_PLANES, "probe_max_blocks": 3}):  # (the last two: k_probe — a wave per task — for every lead, for leads of up to 3 blocks)
        with options(w.dev, **ps):
            sets, hashes, info = run_docs_only(w, progs)
        if ps.get("plane_div"):
            assert info["plane_terms"] > 0
        assert (info["probe_queries"] > 0) == bool(ps.get("probe_max_blocks")) or w.dev.get_option("planes") == 0 or w.dev.get_option("plane_div") == 0, (ps, info["probe_queries"])
        for t, got, want, h in zip(texts, sets, wants, hashes):
            assert np.array_equal(got, want), (ps, t, len(got), len(want))
            assert int(h) == O.fnv1a_docs(want)


def test_candidate_tile_queues_per_xcd(large):
    """k_and draws its tasks from one queue per XCD.  On the 2M-document segment a batch of two- and three-term conjunctions has leads long enough
    for the planner to order the queues by the plane row the tasks probe (`cand_xcd`, the default: planner.hpp "k_and's queues"); without it the
    cost order is dealt round the queues.  Either way every task runs exactly once, whoever draws it: the docID sets equal the oracle's."""
    w, T = large, large.T
    texts = [f"t{a} t{b}" for a, b in T.gen_queries(w.V, 31, 600, 2).tolist()] + [f"t{a} t{b} t{c}" for a, b, c in T.gen_queries(w.V, 32, 100, 3).tolist()]
    progs = [O.parse_query(t) for t in texts]
    wants = [w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)[0] for p in progs]
    for xcd in (1, 0):
        with options(w.dev, cand_xcd=xcd):
            sets, hashes, info = run_docs_only(w, progs)
        assert info["cand_queries"] > 400, info["cand_queries"]
        for t, got, want, h in zip(texts, sets, wants, hashes):
            assert np.array_equal(got, want), (xcd, t, len(got), len(want))
            assert int(h) == O.fnv1a_docs(want)


@pytest.mark.parametrize("k", [3, 5])
def test_and_k_terms_matches_oracle(small, dense, k):
    for w in (small, dense):
        T = w.T
        qs = T.gen_queries(w.V, 99, 150, k).tolist() + [list(range(k)), list(range(k))[::-1]]
        sets, _, _ = run_docs_only(w, [and_prog(T, q) for q in qs])
        for q, got in zip(qs, sets):
            want, _ = w.ora.exec(and_prog(T, q), O.FLAG_DOCUMENTS_ONLY)
            assert np.array_equal(got, want), q


def test_and_edge_cases(small):
    w, T = small, small.T
    V = w.V
    cases = [
        [0, V + 7],  # unknown term: no documents (index_source.h:60-72) => empty
        [0, 0],  # the same term twice
        [5],  # single term == its postings list
        [V - 1, V - 2],  # two rare terms
    ]
    progs = [and_prog(T, c) if len(c) > 1 else np.array([T.tok(T.OP_TERM, c[0])], dtype=np.uint32) for c in cases]
    sets, _, _ = run_docs_only(w, progs)
    assert len(sets[0]) == 0
    assert np.array_equal(sets[1], w.ora.decode_term(0)[0])
    assert np.array_equal(sets[2], w.ora.decode_term(5)[0])
    want, _ = w.ora.exec(and_prog(T, cases[3]), O.FLAG_DOCUMENTS_ONLY)
    assert np.array_equal(sets[3], want)


def test_and_against_reference_fixtures(T, dev):
    """DocumentsOnly conjunction records of tests/golden/ref_*.json (outputs of the genuine reference)."""
    checked = 0
    for name in ("tiny", "small", "dense"):
        g = json.load(open(os.path.join(GOLDEN, f"ref_{name}.json")))
        c = g["corpus"]
        w = World(T, dev, c["D"], c["V"], c["slots"], c["seed"])
        recs = [r for r in g["results"] if r["cmd"] in ("query", "queryfull") and r["flags"] == 1 and gpu_lowers(r["q"])]
        progs = [O.parse_query(r["q"]) for r in recs]
        sets, hashes, _ = run_docs_only(w, progs)
        for r, got, h in zip(recs, sets, hashes):
            assert len(got) == r["n"], r["q"]
            assert str(int(h)) == r["fnv"], r["q"]
            k = min(16, len(got))
            assert got[:k].tolist() == r["first"] and got[len(got) - k :].tolist() == r["last"]
            checked += 1
        w.ix.close()
    assert checked >= 200


def test_and_properties_at_scale(medium):
    """Size-independent properties: A∩A = A; A∩B = B∩A (program order is irrelevant); |A∩B| <= min df; results
    ascending and members of both lists."""
    w, T = medium, medium.T
    pairs = [[0, 1], [0, 3], [2, 1], [0, 50], [7, 9000]]
    progs = [and_prog(T, p) for p in pairs] + [and_prog(T, p[::-1]) for p in pairs] + [and_prog(T, [0, 0])]
    sets, _, _ = run_docs_only(w, progs)
    n = len(pairs)
    for i, p in enumerate(pairs):
        a, b = sets[i], sets[n + i]
        assert np.array_equal(a, b)
        assert len(a) <= min(w.df(p[0]), w.df(p[1]))
        assert np.all(a[1:] > a[:-1])
        da = w.ora.decode_term(p[0])[0]
        db = w.ora.decode_term(p[1])[0]
        assert np.array_equal(a, np.intersect1d(da, db))
    assert np.array_equal(sets[-1], w.ora.decode_term(0)[0])


# ------------------------------------------------------------------------------------------ BM25 + top-K (K5)
SIMS = {"bm25": 0, "tfidf": 1, "trivial": 2}  # TRI_SIM_* == O.SIM_*


def run_scored(w, programs, k, similarity=0):
    b = w.T.Batch(w.ix, programs, w.T.FLAG_ACCUMULATED_SCORE, topk=k, similarity=similarity)
    b.run()
    b.sync()
    d, s, c = b.topk_results()
    counts = b.counts()
    b.close()
    return d, s, c, counts


@pytest.mark.parametrize("world,nq,k", [("small", 200, 10), ("dense", 150, 100), ("medium", 120, 100), ("small", 60, 256)])
def test_scored_and_topk_matches_oracle(request, world, nq, k):
    """AccumulatedScoreScheme + BM25 + application top-K: docIDs exact (score desc, docID asc), scores within 1e-5
    relative (the tolerance BASELINE.json states), total match counts exact."""
    w = request.getfixturevalue(world)
    T = w.T
    qs = T.gen_queries(w.V, 21, nq, 2).tolist() + T.gen_queries(w.V, 22, nq // 4, 3).tolist() + [[0, 1], [1, 0], [0, 1, 2, 3, 4], [0, 0], [5]]
    progs = [and_prog(T, q) if len(q) > 1 else np.array([T.tok(T.OP_TERM, q[0])], dtype=np.uint32) for q in qs]
    d, s, c, counts = run_scored(w, progs, k)
    for i, q in enumerate(qs):
        docs, scores = w.ora.exec(progs[i], O.FLAG_ACCUM_SCORE)
        assert int(counts[i]) == len(docs), q
        td, ts = w.ora.topk(docs, scores, k)
        assert int(c[i]) == len(td), q
        assert d[i, : len(td)].tolist() == td.tolist(), q
        np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0)


def test_scored_against_reference_fixtures(T, dev):
    """flags=2 conjunction records of the reference fixtures: count, top-10 (docID, score)."""
    checked = 0
    for name in ("small", "dense"):
        g = json.load(open(os.path.join(GOLDEN, f"ref_{name}.json")))
        c = g["corpus"]
        w = World(T, dev, c["D"], c["V"], c["slots"], c["seed"])
        for sim, simid in SIMS.items():  # BM25 and the TF-IDF / Trivial scorers of similarity.h, all from the genuine reference
            recs = [r for r in g["results"] if r["cmd"] == "query" and r["flags"] == 2 and "top" in r and gpu_lowers(r["q"]) and r.get("sim", "bm25") == sim]
            assert len(recs) >= 15, sim
            d, s, cnt, counts = run_scored(w, [O.parse_query(r["q"]) for r in recs], 10, similarity=simid)
            for i, r in enumerate(recs):
                assert int(counts[i]) == r["n"], r["q"]
                top = r["top"]
                assert d[i, : len(top)].tolist() == [x[0] for x in top], (sim, r["q"])
                np.testing.assert_allclose(s[i, : len(top)], [x[1] for x in top], rtol=1e-5)
                checked += 1
        w.ix.close()
    assert checked >= 150


# ------------------------------------------------------------------------------------------ OR and mixed AND/OR (K4)
TEMPLATES = ["t{a} OR t{b}", "t{a} OR t{b} OR t{c} OR t{d} OR t{e}", "t{a} t{b} (t{c} OR t{d} OR t{e})", "(t{a} OR t{b}) (t{c} OR t{d}) t{e}", "t{a} (t{b} OR t{c})", "(t{a} OR t{b}) (t{c} OR t{d})"]


def template_queries(w, seed, n):
    rows = w.T.gen_queries(w.V, seed, n, 5).tolist() + [[0, 1, 2, 3, 4], [4, 3, 2, 1, 0], [0, w.V - 1, 1, w.V - 2, 2]]
    out = []
    for i, r in enumerate(rows):
        a, b, c, d, e = r
        for tpl in TEMPLATES:
            out.append(tpl.format(a=a, b=b, c=c, d=d, e=e))
    return out


@pytest.mark.parametrize("world,n", [("small", 40), ("dense", 30), ("medium", 25)])
def test_or_and_mixed_docsets_match_oracle(request, world, n):
    w = request.getfixturevalue(world)
    texts = template_queries(w, 31, n)
    progs = [O.parse_query(t) for t in texts]
    sets, hashes, _ = run_docs_only(w, progs)
    for t, p, got, h in zip(texts, progs, sets, hashes):
        want, _ = w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)
        assert np.array_equal(got, want), (t, len(got), len(want))
        assert int(h) == O.fnv1a_docs(want)


@pytest.mark.parametrize("world,n,k", [("small", 30, 100), ("dense", 20, 10), ("medium", 15, 100)])
def test_or_and_mixed_scored_topk_match_oracle(request, world, n, k):
    w = request.getfixturevalue(world)
    texts = template_queries(w, 32, n)
    progs = [O.parse_query(t) for t in texts]
    d, s, c, counts = run_scored(w, progs, k)
    for i, t in enumerate(texts):
        docs, scores = w.ora.exec(progs[i], O.FLAG_ACCUM_SCORE)
        assert int(counts[i]) == len(docs), t
        td, ts = w.ora.topk(docs, scores, k)
        assert d[i, : len(td)].tolist() == td.tolist(), t
        np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0)


@pytest.mark.parametrize("codec", [1, 2])
def test_plane_rows_are_built_by_need(T, dev, codec):
    """The index's plane cache holds a row in two parts (dev_structs.hpp: PL_HI): plane 0 — all a DocumentsOnly batch reads — is built when any batch's run
    names the term, the high part (nested planes 1 .. 3 + the level words: six times as large) only once a SCORED batch does.  One fresh index,
    DocumentsOnly -> scored (the one-pass kernel and match-then-score) -> DocumentsOnly again, every result checked against the oracle; what each run decoded
    says which parts were built when."""
    w = World(T, dev, 60000, 3000, 10, 11, codec=codec)
    texts = ["t0 t1", "t0 OR t3", "t2 t5 t9", "t1 (t4 OR t7)", "t0 t1 t2 t3 t4", "t6 OR t8 OR t11 OR t40 OR t90", "t3 t5 NOT t1", f"t0 t{w.V - 1}", f"t{w.V // 2} t1"]
    progs = [O.parse_query(t) for t in texts]

    def docs_only():
        sets, _, info = run_docs_only(w, progs)
        for t, p, got in zip(texts, progs, sets):
            want, _ = w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)
            assert np.array_equal(got, want), (t, len(got), len(want))
        return info

    def scored(**opts):
        with options(w.dev, **opts):
            b = T.Batch(w.ix, progs, T.FLAG_ACCUMULATED_SCORE, topk=10)
        b.run()
        b.sync()
        d, s, c = b.topk_results()
        counts = b.counts()
        info = b.info()
        b.close()
        for i, t in enumerate(texts):
            docs, scores = w.ora.exec(progs[i], O.FLAG_ACCUM_SCORE)
            assert int(counts[i]) == len(docs), t
            td, ts = w.ora.topk(docs, scores, 10)
            assert d[i, : len(td)].tolist() == td.tolist(), t
            np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0)
        return info

    try:
        i1 = docs_only()
        assert i1["plane_terms"] >= 5 and i1["term_planes_decoded_bytes"] > 0
        per_row0 = i1["plane_bytes"] // i1["plane_terms"]  # plane 0 alone: a bitmap over the docID space
        assert i1["plane_bytes"] == i1["plane_terms"] * per_row0 and per_row0 >= w.D // 8
        assert docs_only()["term_planes_decoded_bytes"] == 0  # (its rows are there)
        i2 = scored(dense_min_postings=0)  # (every eligible query through the one-pass kernel: k_planes reads the high parts)
        assert i2["planes_queries"] >= 3 and i2["term_planes_decoded_bytes"] > 0 and i2["plane_bytes"] == i2["plane_terms"] * 7 * per_row0
        i3 = scored(fused=0, planes=3)  # (match, then score: k_score reads the level words of the same rows)
        assert i3["planes_queries"] == 0 and i3["plane_bytes"] == i3["plane_terms"] * 7 * per_row0
        assert scored(dense_min_postings=0)["term_planes_decoded_bytes"] == 0
        i4 = docs_only()
        assert i4["term_planes_decoded_bytes"] == 0 and i4["plane_bytes"] == i1["plane_bytes"]
    finally:
        w.ix.close()


def test_union_of_head_terms_large(large):
    w, T = large, large.T
    texts = ["t0 OR t1", "t0 OR t1 OR t2 OR t3 OR t4", "t0 t1 (t2 OR t3 OR t4)", "(t0 OR t1) (t2 OR t3) t4", "t100000 OR t150000 OR t199999"]
    progs = [O.parse_query(t) for t in texts]
    sets, _, _ = run_docs_only(w, progs)
    for t, p, got in zip(texts, progs, sets):
        want, _ = w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)
        assert np.array_equal(got, want), (t, len(got), len(want))


@pytest.mark.parametrize("world", ["large", "medium_l"])
def test_scatter_unions_list_their_rare_terms_once(request, world):
    """PSET_UNIT_SCATTER (k_psets.hpp): a DocumentsOnly union of head terms with terms that have NO plane — k_psets_prep lists those terms' documents once per
    query, task by task (count, scan, place), and every task of k_psets ORs its slice into the words it has just stored.  Terms of a few rows and of hundreds,
    the same rare term in several queries, a batch run twice (the list's cursor restarts), masked documents, a plane threshold that leaves most terms without a
    plane — every set against the oracle, counts and hashes included."""
    w = request.getfixturevalue(world)
    T, V = w.T, w.V
    mid = [t for t in (V // 400, V // 100, V // 40, V // 10, V // 3, V - 1) if w.df(t)]
    texts = [f"t0 OR t{mid[0]} OR t{mid[1]}", "t1 OR t2 OR " + " OR ".join(f"t{t}" for t in mid[2:5]), f"t3 OR t{mid[-1]}", f"t0 OR t1 OR t{mid[0]} OR t{mid[-1]} OR t{mid[2]}",
             "t2 OR " + " OR ".join(f"t{t}" for t in mid), f"t4 OR t5 OR t{mid[1]}"]  # fmt: skip
    progs = [O.parse_query(t) for t in texts]
    masked = np.array(sorted(set(np.random.default_rng(11).integers(1, w.D, w.D // 13).tolist())), dtype=np.uint32)
    try:
        for mk in (None, masked):
            if mk is not None:
                w.ix.set_masked(mk)
                w.ora.set_masked(mk)
            want = [w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)[0] for p in progs]
            for opts in ({}, {"dense_min_postings": 0}, {"plane_div": 64}, {"plane_div": 64, "plane_amortize": 1000}):
                with options(w.dev, **opts):
                    b = T.Batch(w.ix, progs, T.FLAG_DOCUMENTS_ONLY)
                for rep in range(2):
                    b.run()
                    b.sync()
                    counts, hashes = b.counts(), b.docset_hashes()
                    for i, t in enumerate(texts):
                        assert int(counts[i]) == len(want[i]) and int(hashes[i]) == O.fnv1a_docs(want[i]), (opts, rep, t, int(counts[i]), len(want[i]))
                        assert np.array_equal(b.docset(i, len(want[i])), want[i]), (opts, rep, t)
                if not opts and mk is None and b.info()["bitmap_queries"]:  # (TRINITY_TEST_OPTIONS may force result_bitmaps = 0: no scatter unions then)
                    assert b.info()["pset_queries"] >= 1, b.info()  # (the default options send head-term unions with rare terms through k_psets)
                b.close()
    finally:
        w.ix.set_masked(np.zeros(0, np.uint32))
        w.ora.set_masked(np.zeros(0, np.uint32))


@pytest.mark.parametrize("world", ["large", "medium_l"])
def test_docsets_in_one_call(request, world):
    """tri_batch_docsets: every query's ascending docID set, queries in the caller's order, one device-side gather + one copy — conjunctions cut
    into many candidate-tile tasks, unions, phrases, NOT, a general tree and an empty result in one batch, also into a caller's buffer
    and in the default (MatchedTerms) mode; an AccumulatedScore top-K batch refuses."""
    w = request.getfixturevalue(world)
    T, V = w.T, w.V
    texts = [f"t{a} t{b}" for a, b in T.gen_queries(V, 77, 40, 2).tolist()] + ["t0 t1 t2", "t3 OR t5 OR t9", '"t0 t1"', '"t1 t2 t3" t0', "t3 t5 NOT t1", "t2 OR (t0 t1)",
                                                                              f"t{V - 1} t{V - 2} t{V - 3} t{V - 4}", "t0 OR t1 OR t2 OR t3 OR t4"]  # fmt: skip
    progs = [O.parse_query(t) for t in texts]
    want = [w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)[0] for p in progs]
    for flags in (T.FLAG_DOCUMENTS_ONLY, T.FLAG_MATCHED_TERMS):
        with options(w.dev, cand_task_cost=4096):  # (many tasks per query: the gather adds up the earlier tasks' matches)
            b = T.Batch(w.ix, progs, flags)
        b.run()
        b.sync()
        flat, offs = b.docsets()
        assert offs.tolist() == np.concatenate([[0], np.cumsum([len(x) for x in want])]).tolist()
        for i, t in enumerate(texts):
            assert np.array_equal(flat[int(offs[i]) : int(offs[i + 1])], want[i]), t
        mine = np.full(int(offs[-1]) + 7, 0xFFFFFFFF, dtype=np.uint32)  # (a caller's buffer, larger than needed: the tail stays untouched)
        flat2, offs2 = b.docsets(out=mine)
        assert flat2 is mine and np.array_equal(mine[: int(offs[-1])], flat[: int(offs[-1])]) and np.array_equal(offs, offs2) and (mine[int(offs[-1]) :] == 0xFFFFFFFF).all()
        with pytest.raises(T.TrinityError):
            b.docsets(out=np.zeros(max(1, int(offs[-1]) - 1), dtype=np.uint32))  # (too small: refused, nothing written past it)
        b.close()
    sb = T.Batch(w.ix, progs[:4], T.FLAG_ACCUMULATED_SCORE, topk=10)
    sb.run()
    sb.sync()
    if sb.info()["planes_queries"] + sb.info()["fused_queries"]:
        with pytest.raises(T.TrinityError):
            sb.docsets()
    sb.close()


@pytest.mark.parametrize("world", ["large", "medium_l"])
def test_docsets_with_phrase_leaves_of_trees(request, world):
    """tri_batch_docsets over a batch whose TASK_TREE queries have multi-word phrase LEAVES: the planner evaluates such a leaf as a hidden query
    (qid 0xffffffff, no place in the caller's order) whose tasks must not be delivered — round 5 copied their segments to flat[0 ...] over the
    first query's set, and past the buffer's end when the hidden phrase matched more documents than the batch delivers (`t5 NOT "t0 t1"`)."""
    w = request.getfixturevalue(world)
    T, V = w.T, w.V
    texts = ["t7 t9", 't2 OR "t0 t1"', 't5 NOT "t0 t1"', f't{V - 1} NOT "t0 t1"', '"t0 t1" OR "t1 t2"', "t3 t4", '(t2 OR "t1 t0") NOT t3']
    progs = [O.parse_query(t) for t in texts]
    want = [w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)[0] for p in progs]
    for flags in (T.FLAG_DOCUMENTS_ONLY, T.FLAG_MATCHED_TERMS):
        b = T.Batch(w.ix, progs, flags)
        b.run()
        b.sync()
        assert b.info()["tree_queries"] >= 4
        for rep in range(2):
            flat, offs = b.docsets()
            assert offs.tolist() == np.concatenate([[0], np.cumsum([len(x) for x in want])]).tolist()
            for i, t in enumerate(texts):
                assert np.array_equal(flat[int(offs[i]) : int(offs[i + 1])], want[i]), t
        b.close()


@pytest.mark.parametrize("world", ["large", "dense", "medium_l"])
def test_docsets_delivered_as_bitmaps(request, world):
    """RESULT_BITMAP (dev_structs.hpp): a DocumentsOnly union / conjunction of head terms expected to match one document in 32 or more is held
    as one bit per document — through both bitmap-window kernels (k_psets: every term has a plane; k_and_dense: rows decoded, windows the lead
    group skips written as zeros), with and without masked documents.  tri_batch_docset expands it, tri_batch_docset_bitmap hands the words
    over, counts and hashes read the same; result_bitmaps = 0 gives the same sets as ascending docIDs."""
    w = request.getfixturevalue(world)
    T, V = w.T, w.V
    texts = ["t0 OR t1", "t0 OR t1 OR t2 OR t3 OR t4", "t0 t1", "t0 t1 (t2 OR t3 OR t4)", "(t0 OR t1) (t2 OR t3) t4", "(t0 OR t1 OR t2) NOT t3", f"t0 OR t{V // 2} OR t{V - 1}",
             f"t{V // 2} OR t{V // 2 + 1}", "t5 t6", f"(t0 OR t{V - 2}) (t1 OR t{V - 3})"]  # fmt: skip
    progs = [O.parse_query(t) for t in texts]
    masked = np.array(sorted(set(np.random.default_rng(5).integers(1, w.D, w.D // 9).tolist())), dtype=np.uint32)
    try:
        for mk in (None, masked):
            if mk is not None:
                w.ix.set_masked(mk)
                w.ora.set_masked(mk)
            want = [w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)[0] for p in progs]
            seen_forms = set()
            for opts in ({}, {"dense_min_postings": 0}, {"dense_min_postings": 0, "planes": 0}, {"dense_min_postings": 0, "plane_div": ALL_PLANES}, {"result_bitmaps": 0}):
                with options(w.dev, **opts):
                    b = T.Batch(w.ix, progs, T.FLAG_DOCUMENTS_ONLY)
                for rep in range(2):
                    b.run()
                    b.sync()
                    info = b.info()
                    assert (info["bitmap_queries"] == 0) == (opts.get("result_bitmaps", 1) == 0 or info["dense_queries"] + info["pset_queries"] == 0), opts
                    counts, hashes = b.counts(), b.docset_hashes()
                    nbm = 0
                    for i, t in enumerate(texts):
                        assert int(counts[i]) == len(want[i]) and int(hashes[i]) == O.fnv1a_docs(want[i]), (opts, t)
                        assert np.array_equal(b.docset(i, len(want[i])), want[i]), (opts, t)
                        bm = b.docset_bitmap(i)
                        if bm is not None:
                            first, words = bm
                            bits = np.unpackbits(words.view(np.uint8), bitorder="little")
                            assert first % 32 == 0 and np.array_equal(np.nonzero(bits)[0].astype(np.uint32) + first, want[i]), (opts, t)
                            nbm += 1
                    assert nbm == info["bitmap_queries"]
                    seen_forms.add(nbm > 0)
                    flat, offs = b.docsets()  # every set in one call: both forms through k_deliver_docsets
                    assert int(offs[-1]) == sum(len(x) for x in want)
                    for i, t in enumerate(texts):
                        assert np.array_equal(flat[int(offs[i]) : int(offs[i + 1])], want[i]), (opts, t)
                    # ... and each set in the form the engine holds it (tri_batch_docsets_mixed): the dense ones as the words of their bitmaps
                    mflat, moffs, forms = b.docsets_mixed()
                    assert int(forms.sum()) == nbm
                    for i, t in enumerate(texts):
                        part = mflat[int(moffs[i]) : int(moffs[i + 1])]
                        if forms[i]:
                            bits = np.unpackbits(part.view(np.uint8), bitorder="little")
                            assert np.array_equal(np.nonzero(bits)[0].astype(np.uint32), want[i]), (opts, t)
                        else:
                            assert np.array_equal(part, want[i]), (opts, t)
                    if nbm:
                        assert int(moffs[-1]) < int(offs[-1])  # (a dense set weighs less as bits than as docIDs: that is when the planner chooses the form)
                b.close()
            assert seen_forms == {True, False}
    finally:
        w.ix.set_masked(np.zeros(0, np.uint32))
        w.ora.set_masked(np.zeros(0, np.uint32))


# ------------------------------------------------------------------------------------------ one-pass scored windows (k_fused)
FUSED_EXTRA = ["t{a} OR t{b} OR t{c} OR t{d} OR t{e} OR t0 OR t1 OR t2", "t{a} t{b} (t{c} OR t{d} OR t{e} OR t0 OR t1)", "t{a}", "t{a} t{b}", "t{a} t{a}", "t{a} t{b} t{c} t{d} t{e}", "t{a} NOT t{b}", "(t{a} OR t{b}) NOT t{c}", "t{a} t{b} NOT (t{c} OR t{d})",
               "t{a} <t{b}>", "t{a} t{b} <t{c} OR t{d}>", "(t{a} OR t{b}) (t{a} OR t{c})", "t{a} OR t{b} OR t{c} OR t{d} OR t{e} OR t{a}"]


def fused_queries(w, seed, n):
    rows = w.T.gen_queries(w.V, seed, n, 5).tolist() + [[0, 1, 2, 3, 4], [4, 3, 2, 1, 0], [0, w.V - 1, 1, w.V - 2, 2]]
    return [tpl.format(a=a, b=b, c=c, d=d, e=e) for a, b, c, d, e in rows for tpl in TEMPLATES + FUSED_EXTRA]


def check_scored(w, texts, progs, k, similarity=0, tag=None):
    d, s, c, counts = run_scored(w, progs, k, similarity=similarity)
    for i, t in enumerate(texts):
        docs, scores = w.ora.exec(progs[i], O.FLAG_ACCUM_SCORE)
        assert int(counts[i]) == len(docs), (tag, t)
        td, ts = w.ora.topk(docs, scores, k)
        assert int(c[i]) == len(td), (tag, t)
        assert d[i, : len(td)].tolist() == td.tolist(), (tag, t)
        np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0)


@pytest.mark.parametrize("world,n,k", [("small", 12, 10), ("dense", 10, 100), ("small_l", 12, 100), ("dense_l", 10, 256), ("medium", 5, 100), ("medium_l", 6, 10)])
def test_fused_scored_windows_match_oracle(request, world, n, k):
    """AccumulatedScore top-K with every eligible query forced through the one-pass kernel (dense_min_postings = 0): conjunctions,
    unions, CNF, NOT, optional terms, repeated terms, single terms; both codecs; then with the window fields saturating at freq 1
    and 3 (every document above is rescored from the postings) and, as the cross-check, with the one-pass kernel off."""
    w = request.getfixturevalue(world)
    texts = fused_queries(w, 41, n)
    progs = [O.parse_query(t) for t in texts]
    # k_planes (bit planes): the planner's term planes, a plane for every term, none (every slot decoded per window), small tasks;
    # then k_fused (window words, planes off) in its variants; then match-then-score
    for opts in ({"dense_min_postings": 0}, {"dense_min_postings": 0, "plane_div": ALL_PLANES}, {"dense_min_postings": 0, "plane_div": 0},
                 {"dense_min_postings": 0, "planes_split": 1 << 20, "fused_task_cost": 4096}, {"dense_min_postings": 0, "planes_split": 1 << 20, "fused_task_cost": 4096, "plane_div": 0},
                 {"dense_min_postings": 0, "planes_split": 1}, {"dense_min_postings": 0, "planes_split": 7}, {"dense_min_postings": 0, "planes_split": 64, "plane_div": 0},
                 {"dense_min_postings": 0, "planes": 0}, {"dense_min_postings": 0, "planes": 0, "fused_freq_cap": 1}, {"dense_min_postings": 0, "planes": 0, "fused_freq_cap": 3},
                 {"dense_min_postings": 0, "planes": 0, "fused_task_cost": 4096}, {"dense_min_postings": 0, "planes": 0, "fused_halfwords": 0},
                 {"dense_min_postings": 0, "planes": 0, "fused_halfwords": 0, "fused_freq_cap": 2}, {"dense_min_postings": 0, "fused": 0}):
        with options(w.dev, **opts):
            check_scored(w, texts, progs, k, tag=opts)


@pytest.mark.parametrize("sim", ["tfidf", "trivial"])
@pytest.mark.parametrize("world", ["dense", "dense_l"])
def test_fused_other_similarities(request, world, sim):
    w = request.getfixturevalue(world)
    texts = fused_queries(w, 43, 6)
    progs = [O.parse_query(t) for t in texts]
    w.ora.set_similarity(SIMS[sim])
    try:
        for opts in ({"dense_min_postings": 0}, {"dense_min_postings": 0, "plane_div": ALL_PLANES}, {"dense_min_postings": 0, "plane_div": 0},
                     {"dense_min_postings": 0, "planes": 0}, {"dense_min_postings": 0, "planes": 0, "fused_freq_cap": 2}):
            with options(w.dev, **opts):
                check_scored(w, texts, progs, 50, similarity=SIMS[sim], tag=(sim, opts))
    finally:
        w.ora.set_similarity(0)


def test_fused_large_unions_and_cnf(large):
    """2M documents: unions and CNFs of head terms (millions of matches per query, hundreds of windows, several tasks per query)."""
    w = large
    texts = ["t0 OR t1", "t0 OR t1 OR t2 OR t3 OR t4", "t0 t1 (t2 OR t3 OR t4)", "(t0 OR t1) (t2 OR t3) t4", "t0 t1", "t0 t1 t2 t3 t4", "t100000 OR t150000 OR t199999",
             "t0 OR t199999", "t3 t5 NOT t1", "t2 <t7 OR t9>"]
    progs = [O.parse_query(t) for t in texts]
    # (k_planes: a query's docID ranges — one task each — share its threshold; 1 range, the default 2, many, and the cut by postings)
    for opts in ({}, {"planes_split": 1}, {"planes_split": 16}, {"planes_split": 1 << 20, "fused_task_cost": 200000}, {"plane_div": 0}, {"plane_div": 0, "planes_split": 1 << 20, "fused_task_cost": 200000},
                 {"planes": 0}, {"planes": 0, "fused_task_cost": 200000}):
        with options(w.dev, **opts):
            check_scored(w, texts, progs, 100, tag=opts)


def test_fused_batches_do_not_materialise_docsets(small):
    """An AccumulatedScore top-K batch run through the one-pass kernel keeps top-K lists and counts; asking it for a docID set
    fails with TRI_ERR_INVALID instead of returning stale memory."""
    w, T = small, small.T
    with options(w.dev, dense_min_postings=0):
        b = T.Batch(w.ix, [O.parse_query("t0 OR t1")], T.FLAG_ACCUMULATED_SCORE, topk=10)
    b.run()
    b.sync()
    assert int(b.counts()[0]) > 0 and b.info()["fused_queries"] + b.info()["planes_queries"] == 1
    with pytest.raises(T.TrinityError):
        b.docset(0)
    with pytest.raises(T.TrinityError):
        b.docset_hashes()
    b.close()


# ------------------------------------------------------------------------------------------ phrases (K6)
PHRASE_TEMPLATES = ['"t{a} t{b}"', '"t{a} t{b} t{c}"', '"t{a} t{b}" t{c}', '"t{a} t{b}" "t{c} t{d}"', '"t{a} t{a}"', '"t{a} t{b} t{a}"', 't{e} "t{b} t{a}"',
                    # phrases that share terms (one DocWordsSpace / one set of term hits per candidate document, queryexec_ctx.cpp:317-351), a
                    # phrase next to one of its own terms
                    '"t{a} t{b}" "t{b} t{c}"', '"t{a} t{b}" "t{c} t{a}"', '"t{a} t{b} t{c}" "t{b} t{c}"', '"t{a} t{b}" t{a}']


def phrase_queries(w, seed, n):
    rows = w.T.gen_queries(w.V, seed, n, 5).tolist()
    head = [[0, 1, 2, 3, 4], [1, 0, 2, 5, 3], [2, 0, 1, 4, 7], [0, 2, 1, 3, 5], [3, 1, 0, 2, 6], [1, 2, 0, 4, 3]]
    out = []
    for r in head + rows:
        a, b, c, d, e = r
        for tpl in PHRASE_TEMPLATES:
            out.append(tpl.format(a=a, b=b, c=c, d=d, e=e))
    return out


@pytest.mark.parametrize("world,n", [("small", 25), ("dense", 25), ("medium", 10), ("longdocs", 12)])
def test_phrase_docsets_match_oracle(request, world, n):
    w = request.getfixturevalue(world)
    texts = phrase_queries(w, 41, n)
    progs = [O.parse_query(t) for t in texts]
    sets, hashes, _ = run_docs_only(w, progs)
    nonempty = 0
    for t, p, got, h in zip(texts, progs, sets, hashes):
        want, _ = w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)
        assert np.array_equal(got, want), (t, len(got), len(want))
        assert int(h) == O.fnv1a_docs(want)
        nonempty += len(want) > 0
    assert nonempty >= 20


@pytest.mark.parametrize("world,n,k", [("small", 15, 10), ("dense", 15, 100), ("longdocs", 10, 100)])
def test_phrase_scored_topk_match_oracle(request, world, n, k):
    """Phrase scoring: scorer->score(id, matchCnt, sum of the terms' idf) — matchCnt counts every start position in
    AccumulatedScoreScheme (docset_iterators_scorers.cpp:195-228, exec.cpp:296)."""
    w = request.getfixturevalue(world)
    texts = phrase_queries(w, 42, n)
    progs = [O.parse_query(t) for t in texts]
    d, s, c, counts = run_scored(w, progs, k)
    for i, t in enumerate(texts):
        docs, scores = w.ora.exec(progs[i], O.FLAG_ACCUM_SCORE)
        assert int(counts[i]) == len(docs), t
        td, ts = w.ora.topk(docs, scores, k)
        assert d[i, : len(td)].tolist() == td.tolist(), t
        np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0)


# ------------------------------------------------------------------------------------------ LUCENE-shaped codec (K2)
@pytest.fixture(scope="module")
def small_l(T, dev):
    return World(T, dev, 20000, 2000, 10, 42, codec=2)


@pytest.fixture(scope="module")
def dense_l(T, dev):
    return World(T, dev, 20000, 500, 12, 7, codec=2)


@pytest.fixture(scope="module")
def medium_l(T, dev):
    return World(T, dev, 300000, 30000, 10, 42, codec=2)


@pytest.mark.parametrize("world", ["small_l", "dense_l", "medium_l"])
def test_lucene_decode_terms_bit_exact(request, world):
    w = request.getfixturevalue(world)
    terms = [t for t in [0, 1, 2, 3, 5, 17, 40, w.V // 3, w.V // 2, w.V - 1] if w.df(t)]
    docs, freqs, offs = w.ix.decode_terms(terms, [w.df(t) for t in terms])
    for i, t in enumerate(terms):
        d, f = w.ora.decode_term(t)
        assert np.array_equal(docs[offs[i] : offs[i + 1]], d), t
        assert np.array_equal(freqs[offs[i] : offs[i + 1]], f), t


@pytest.mark.parametrize("world,n", [("small_l", 40), ("dense_l", 30), ("medium_l", 25)])
def test_lucene_docsets_match_oracle(request, world, n):
    w = request.getfixturevalue(world)
    T = w.T
    texts = template_queries(w, 51, n) + [f"t{a} t{b}" for a, b in T.gen_queries(w.V, 52, 150, 2).tolist()] + ["t0 t1", "t0 t1 t2 t3 t4", "t5"]
    progs = [O.parse_query(t) for t in texts]
    sets, hashes, _ = run_docs_only(w, progs)
    for t, p, got, h in zip(texts, progs, sets, hashes):
        want, _ = w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)
        assert np.array_equal(got, want), (t, len(got), len(want))
        assert int(h) == O.fnv1a_docs(want)


@pytest.mark.parametrize("world,n,k", [("small_l", 25, 100), ("dense_l", 20, 10)])
def test_lucene_scored_topk_match_oracle(request, world, n, k):
    """cfg3's shape: 5-term mixed AND/OR, BM25, top-K over the Lucene-shaped codec."""
    w = request.getfixturevalue(world)
    texts = template_queries(w, 53, n) + ["t0 t1", "t0 t1 t2 t3 t4"]
    progs = [O.parse_query(t) for t in texts]
    d, s, c, counts = run_scored(w, progs, k)
    for i, t in enumerate(texts):
        docs, scores = w.ora.exec(progs[i], O.FLAG_ACCUM_SCORE)
        assert int(counts[i]) == len(docs), t
        td, ts = w.ora.topk(docs, scores, k)
        assert d[i, : len(td)].tolist() == td.tolist(), t
        np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0)


@pytest.mark.parametrize("world,n", [("small_l", 25), ("dense_l", 25), ("medium_l", 10)])
def test_lucene_phrase_docsets_match_oracle(request, world, n):
    """a10: positions from hits.data (128-hit blocks independent of the document blocks + varbyte tail)."""
    w = request.getfixturevalue(world)
    texts = phrase_queries(w, 61, n)
    progs = [O.parse_query(t) for t in texts]
    sets, hashes, _ = run_docs_only(w, progs)
    nonempty = 0
    for t, p, got, h in zip(texts, progs, sets, hashes):
        want, _ = w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)
        assert np.array_equal(got, want), (t, len(got), len(want))
        assert int(h) == O.fnv1a_docs(want)
        nonempty += len(want) > 0
    assert nonempty >= 20


@pytest.mark.parametrize("world,n,k", [("small_l", 15, 10), ("dense_l", 15, 100)])
def test_lucene_phrase_scored_topk_match_oracle(request, world, n, k):
    w = request.getfixturevalue(world)
    texts = phrase_queries(w, 62, n)
    progs = [O.parse_query(t) for t in texts]
    d, s, c, counts = run_scored(w, progs, k)
    for i, t in enumerate(texts):
        docs, scores = w.ora.exec(progs[i], O.FLAG_ACCUM_SCORE)
        assert int(counts[i]) == len(docs), t
        td, ts = w.ora.topk(docs, scores, k)
        assert d[i, : len(td)].tolist() == td.tolist(), t
        np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0)


def test_lucene_phrase_fixtures(T, dev):
    """The reference's own phrase results (produced through its Google codec) from a LUCENE-coded segment of the same corpus."""
    checked = 0
    for name in ("small", "dense"):
        g = json.load(open(os.path.join(GOLDEN, f"ref_{name}.json")))
        c = g["corpus"]
        w = World(T, dev, c["D"], c["V"], c["slots"], c["seed"], codec=2)
        recs = [r for r in g["results"] if r["cmd"] in ("query", "queryfull") and r["flags"] == 1 and '"' in r["q"]]
        sets, hashes, _ = run_docs_only(w, [O.parse_query(r["q"]) for r in recs])
        for r, got, h in zip(recs, sets, hashes):
            assert len(got) == r["n"] and str(int(h)) == r["fnv"], r["q"]
            checked += 1
        w.ix.close()
    assert checked >= 30


def test_lucene_phrase_needs_hits(T, dev):
    seg = T.Segment(2000, 200, 10, 42, codec=2)
    ix = T.Index(dev, seg.index, seg.terms, seg.docs_cnt, codec=2)  # no hits.data
    with pytest.raises(T.TrinityError):
        T.Batch(ix, [O.parse_query('"t0 t1"')], T.FLAG_DOCUMENTS_ONLY)
    ix.close()


def test_lucene_forced_dense_and_fixtures(T, dev):
    """Reference fixture records (non-phrase) against a LUCENE-coded segment, bitmap-window path forced."""
    checked = 0
    for name in ("small", "dense"):
        g = json.load(open(os.path.join(GOLDEN, f"ref_{name}.json")))
        c = g["corpus"]
        w = World(T, dev, c["D"], c["V"], c["slots"], c["seed"], codec=2)
        recs = [r for r in g["results"] if r["cmd"] in ("query", "queryfull") and r["flags"] == 1 and '"' not in r["q"] and gpu_lowers(r["q"])]
        with options(dev, dense_min_postings=0):
            sets, hashes, _ = run_docs_only(w, [O.parse_query(r["q"]) for r in recs])
        for r, got, h in zip(recs, sets, hashes):
            assert len(got) == r["n"] and str(int(h)) == r["fnv"], r["q"]
            checked += 1
        w.ix.close()
    assert checked >= 100


# ------------------------------------------------------------------------------------------ SURVEY §8(d) workloads (bench.py --workload)
@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
def test_workload_matches_oracle(T, dev, name):
    """The query sets bench.py times, at a size the oracle finishes in seconds: same programs, bit-exact docID sets
    (scored sets: top-K equal, BM25 within 1e-5).  cfg4's document-sampled phrases must match somewhere."""
    from trinity_amd import workloads as W

    D, V = 30000, 3000
    parts, _ = W.build_parts(name, D, V, 10, 42, 160)
    for progs, flags, topk, codec in ((pt.programs, pt.flags, pt.topk, pt.codec) for pt in parts):
        w = World(T, dev, D, V, 10, 42, codec=codec)
        if flags & T.FLAG_ACCUMULATED_SCORE:
            for opts in ({}, {"dense_min_postings": 0}):  # the planner's choice at this size, then the one-pass scored windows forced
                with options(dev, **opts):
                    check_scored(w, [str(i) for i in range(len(progs))], progs, topk, tag=(name, opts))
        else:
            sets, hashes, _ = run_docs_only(w, progs)
            sampled_hits = 0
            for i, (p, got, h) in enumerate(zip(progs, sets, hashes)):
                want, _ = w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)
                assert np.array_equal(got, want), (name, i, p.tolist(), len(got), len(want))
                assert int(h) == O.fnv1a_docs(want)
                sampled_hits += len(want) > 0
            if name == "cfg4":
                assert sampled_hits >= len(progs) // 2  # every document-sampled phrase occurs in its document
        w.ix.close()


# ------------------------------------------------------------------------------------------ logicalnot (DocsSetIterators::Filter)
NOT_TEMPLATES = ["t{a} NOT t{b}", "t{a} t{b} NOT t{c}", "(t{a} OR t{b}) NOT t{c}", "t{a} NOT (t{b} OR t{c})", "(t{a} OR t{b}) t{d} NOT t{c}",
                 "t{a} (t{b} NOT t{c}) t{d}", "t{a} NOT t{b} NOT t{c}", "t{a} t{b} t{c} NOT (t{d} OR t{e})", "t{a} NOT t{a}", '"t{a} t{b}" NOT t{c}']


def not_queries(w, seed, n):
    rows = w.T.gen_queries(w.V, seed, n, 5).tolist()
    head = [[0, 1, 2, 3, 4], [1, 0, 2, 5, 3], [3, 0, 1, 4, 7], [4, 2, 0, 3, 5], [2, 1, 0, 6, 3], [0, 5, 9, 1, 2]]
    return [tpl.format(a=a, b=b, c=c, d=d, e=e) for a, b, c, d, e in head + rows for tpl in NOT_TEMPLATES]


@pytest.mark.parametrize("world,n", [("small", 20), ("dense", 20), ("medium", 10), ("small_l", 10)])
def test_not_docsets_match_oracle(request, world, n):
    """A -B == docs(A) minus docs(B): both matching kernels (bitmap windows: A & ~B; candidate tiles: keep the not-hit)."""
    w = request.getfixturevalue(world)
    texts = not_queries(w, 71, n)
    progs = [O.parse_query(t) for t in texts]
    sets, hashes, _ = run_docs_only(w, progs)
    for t, p, got, h in zip(texts, progs, sets, hashes):
        want, _ = w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)
        assert np.array_equal(got, want), (t, len(got), len(want))
        assert int(h) == O.fnv1a_docs(want)


def test_not_forced_dense(T, dev):
    w = World(T, dev, 20000, 500, 12, 7)
    texts = not_queries(w, 72, 10)
    progs = [O.parse_query(t) for t in texts]
    with options(dev, dense_min_postings=0):
        sets, _, info = run_docs_only(w, progs)
    for t, p, got in zip(texts, progs, sets):
        want, _ = w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)
        assert np.array_equal(got, want), (t, len(got), len(want))
    w.ix.close()


@pytest.mark.parametrize("world,k", [("small", 10), ("dense", 100)])
def test_not_scored_topk_match_oracle(request, world, k):
    """The excluded side contributes nothing to the score (docset_iterators_scorers.cpp:59-73)."""
    w = request.getfixturevalue(world)
    texts = [t for t in not_queries(w, 73, 8) if '"' not in t]
    progs = [O.parse_query(t) for t in texts]
    d, s, c, counts = run_scored(w, progs, k)
    for i, t in enumerate(texts):
        docs, scores = w.ora.exec(progs[i], O.FLAG_ACCUM_SCORE)
        assert int(counts[i]) == len(docs), t
        td, ts = w.ora.topk(docs, scores, k)
        assert d[i, : len(td)].tolist() == td.tolist(), t
        np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0)


# ------------------------------------------------------------------------------------------ general trees: truth tables in k_fused
TREE_TEMPLATES = [("t{a} NOT (t{b} t{c})", 1), ("t{a} OR (t{b} NOT t{c})", 1), ("t{a} OR (t{b} t{c})", 1), ("(t{a} t{b}) OR (t{c} t{d})", 1), ("t{a} <t{b} t{c}>", 1),
                  ("t{a} (t{b} OR (t{c} t{d}))", 1), ("t{a} NOT (t{b} OR (t{c} t{d}))", 1), ("(t{a} NOT t{b}) OR (t{c} NOT t{d}) OR t{e}", 1), ("t{a} <t{b} NOT t{c}>", 1),
                  ("[t{a}, t{b}, t{c}]", 2), ("[t{a}, t{b}, t{c}, t{d}, t{e}]", 2), ("[t{a}, t{b}, t{c}, t{d}, t{e}]", 3), ("[t{a}, t{b}, t{c}, t{d}, t{e}]", 5), ("[t{a}, t{b}, t{c}, t{d}, t{e}]", 1),
                  ("t{a} [t{b}, t{c}, t{d}]", 2), ("[t{a}, t{b} t{c}, t{d} OR t{e}]", 2), ("t{e} OR [t{a} t{b}, t{c}, t{d} NOT t{a}]", 2), ("[t{a}, t{b}, t{c}] NOT t{d}", 2)]


def tree_queries(w, seed, n):
    rows = w.T.gen_queries(w.V, seed, n, 5).tolist() + [[0, 1, 2, 3, 4], [4, 3, 2, 1, 0], [1, 0, 3, 2, 5], [0, w.V - 1, 1, w.V - 2, 2]]
    texts, progs = [], []
    for a, b, c, d, e in rows:
        for tpl, mn in TREE_TEMPLATES:
            texts.append(f"{tpl.format(a=a, b=b, c=c, d=d, e=e)} /{mn}")
            progs.append(O.parse_query(tpl.format(a=a, b=b, c=c, d=d, e=e), some_min=mn))
    return texts, progs


@pytest.mark.parametrize("world,n", [("small", 8), ("dense", 8), ("dense_l", 6), ("medium", 4)])
def test_general_trees_docsets_match_oracle(request, world, n):
    """matchsome, NOT / Optional of a subtree, AND under OR: DocumentsOnly through the truth-table predicate of the one-pass kernel,
    docID sets equal to the oracle's iterator trees (pinned to the genuine reference on these shapes: tests/golden querysome + NOT records);
    32-bit and 16-bit window words, many small tasks."""
    w = request.getfixturevalue(world)
    texts, progs = tree_queries(w, 71, n)
    for opts in ({}, {"fused_halfwords": 0}, {"fused_task_cost": 2048}):
        with options(w.dev, **opts):
            sets, hashes, info = run_docs_only(w, progs)
        assert info["fused_queries"] > 0
        for t, p, got, h in zip(texts, progs, sets, hashes):
            want, _ = w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)
            assert np.array_equal(got, want), (opts, t, len(got), len(want))
            assert int(h) == O.fnv1a_docs(want), (opts, t)


@pytest.mark.parametrize("world,n,k", [("small", 8, 10), ("dense", 8, 100), ("dense_l", 6, 100), ("medium", 4, 256)])
def test_general_trees_scored_topk_match_oracle(request, world, n, k):
    """... and AccumulatedScore top-K: a document's score is the sum over the scorer leaves that sit on it THROUGH the tree
    (docset_iterators_scorers.cpp:38-57, 77-104, 107-193), which the planner tabulates per presence pattern."""
    w = request.getfixturevalue(world)
    texts, progs = tree_queries(w, 72, n)
    for opts in ({}, {"fused_halfwords": 0}, {"fused_freq_cap": 1}, {"fused_task_cost": 2048}):
        with options(w.dev, **opts):
            check_scored(w, texts, progs, k, tag=opts)


def test_matchsome_against_reference_fixtures(T, dev):
    """The genuine reference's DisjunctionSome answers (`querysome` records: DocumentsOnly sets, AccumulatedScore top-10)."""
    checked = 0
    for name in ("tiny", "small", "dense"):
        g = json.load(open(os.path.join(GOLDEN, f"ref_{name}.json")))
        c = g["corpus"]
        w = World(T, dev, c["D"], c["V"], c["slots"], c["seed"])
        recs = [r for r in g["results"] if r["cmd"] == "querysome" and r["flags"] == 1]
        sets, hashes, _ = run_docs_only(w, [O.parse_query(r["q"], some_min=r["min"]) for r in recs])
        for r, got, h in zip(recs, sets, hashes):
            assert len(got) == r["n"] and str(int(h)) == r["fnv"], (name, r["q"], r["min"])
            checked += 1
        recs = [r for r in g["results"] if r["cmd"] == "querysome" and r["flags"] == 2 and "top" in r]
        d, s, cnt, counts = run_scored(w, [O.parse_query(r["q"], some_min=r["min"]) for r in recs], 10)
        for i, r in enumerate(recs):
            assert int(counts[i]) == r["n"], (name, r["q"], r["min"])
            top = r["top"]
            assert d[i, : len(top)].tolist() == [x[0] for x in top], (name, r["q"], r["min"])
            np.testing.assert_allclose(s[i, : len(top)], [x[1] for x in top], rtol=1e-5)
            checked += 1
        w.ix.close()
    assert checked >= 300


@pytest.mark.parametrize("world,n", [("small", 6), ("dense", 6), ("dense_l", 4)])
def test_general_trees_full_scores_and_rich_mode(request, world, n):
    """... the full score stream (topk == 0: what consider(id, score) receives for every match) and exec_query's default mode: the
    terms reported for a match are those whose iterators sit on it THROUGH the tree (queryexec_ctx.cpp:382-520), not every term it holds."""
    w = request.getfixturevalue(world)
    texts, progs = tree_queries(w, 73, n)
    b = w.T.Batch(w.ix, progs, w.T.FLAG_ACCUMULATED_SCORE, topk=0)
    b.run()
    b.sync()
    counts = b.counts()
    for i, (t, p) in enumerate(zip(texts, progs)):
        docs, scores = w.ora.exec(p, O.FLAG_ACCUM_SCORE)
        assert int(counts[i]) == len(docs), t
        assert np.array_equal(b.docset(i, len(docs)), docs), t
        np.testing.assert_allclose(b.scores(i, len(docs)), scores, rtol=1e-5, atol=0, err_msg=t)
    b.close()
    for t, p, (docs, terms, present, freq, pos) in zip(texts, progs, run_rich(w, progs)):
        wdocs, wflat, tt, ht = w.ora.exec_rich(p)
        assert np.array_equal(docs, wdocs), t
        got = rich_flat(docs, terms, present, freq, pos)
        assert int(freq.sum()) == ht and int(sum(bin(int(x)).count("1") for x in present)) == tt, t
        assert np.array_equal(got, wflat), t


def test_matchsome_rich_mode_against_reference_fixtures(T, dev):
    """`querysome 0` records: the genuine reference's matched terms and hits for DisjunctionSome trees."""
    checked = 0
    for name in ("small", "dense"):
        g = json.load(open(os.path.join(GOLDEN, f"ref_{name}.json")))
        c = g["corpus"]
        w = World(T, dev, c["D"], c["V"], c["slots"], c["seed"])
        recs = [r for r in g["results"] if r["cmd"] == "querysome" and r["flags"] == 0] + [r for r in g["results"] if r["cmd"] == "query" and r["flags"] == 0 and not gpu_lowers(r["q"], rich=True)]
        progs = [O.parse_query(r["q"], some_min=r.get("min", 1)) for r in recs]
        for r, (docs, terms, present, freq, pos) in zip(recs, run_rich(w, progs)):
            assert len(docs) == r["n"] and str(O.fnv1a_docs(docs)) == r["fnv"], r["q"]
            assert int(freq.sum()) == r["hits_total"], r["q"]
            assert str(O.fnv1a_u32_stream(rich_flat(docs, terms, present, freq, pos))) == r["rich_fnv"], r["q"]
            checked += 1
        w.ix.close()
    assert checked >= 80


def test_compiled_exec_trees_through_the_c_abi(T, dev):
    """The reference's own compile_query output (tests/golden/ref_trees.json: exec_node trees + the reference's answers), lowered to
    postfix programs and run on the GPU: DocumentsOnly sets and AccumulatedScore top-10."""
    g = json.load(open(os.path.join(GOLDEN, "ref_trees.json")))
    c = g["corpus"]
    w = World(T, dev, c["D"], c["V"], c["slots"], c["seed"])
    recs = g["results"]
    progs = [np.array(O.program_from_exec_tree(r["tree"]), dtype=np.uint32) for r in recs]
    sets, hashes, _ = run_docs_only(w, progs)
    d, s, cnt, counts = run_scored(w, progs, 10)
    for i, r in enumerate(recs):
        assert len(sets[i]) == r["n"] and str(int(hashes[i])) == r["fnv"], r["q"]
        assert int(counts[i]) == r["n"], r["q"]
        top = r["top"]
        assert d[i, : len(top)].tolist() == [x[0] for x in top], r["q"]
        np.testing.assert_allclose(s[i, : len(top)], [x[1] for x in top], rtol=1e-5)
    w.ix.close()
    assert len(recs) >= 150


def test_random_trees_from_the_reference(T, dev):
    """tests/golden/ref_random.json (384 random trees compiled and answered by the genuine reference) through the C-ABI, all three modes;
    with the CNF-shaped ones also forced through the one-pass kernel."""
    g = json.load(open(os.path.join(GOLDEN, "ref_random.json")))
    c = g["corpus"]
    w = World(T, dev, c["D"], c["V"], c["slots"], c["seed"])
    recs = g["results"]
    progs = [np.array(O.program_from_exec_tree(r["tree"]), dtype=np.uint32) for r in recs]
    for opts in ({}, {"dense_min_postings": 0}):
        with options(dev, **opts):
            sets, hashes, info = run_docs_only(w, progs)
            d, s, cnt, counts = run_scored(w, progs, 10)
        assert info["fused_queries"] > 50  # (the general trees; CNF-shaped ones take the other kernels in DocumentsOnly mode)
        for i, r in enumerate(recs):
            assert len(sets[i]) == r["n"] and str(int(hashes[i])) == r["fnv"], (opts, r["q"])
            assert int(counts[i]) == r["n"], (opts, r["q"])
            top = r["top"]
            assert d[i, : len(top)].tolist() == [x[0] for x in top], (opts, r["q"])
            np.testing.assert_allclose(s[i, : len(top)], [x[1] for x in top], rtol=1e-5)
    for r, (docs, terms, present, freq, pos) in zip(recs, run_rich(w, progs)):
        assert len(docs) == r["n"] and int(freq.sum()) == r["hits_total"], r["q"]
        assert int(sum(bin(int(x)).count("1") for x in present)) == r["terms_total"], r["q"]
        assert str(O.fnv1a_u32_stream(rich_flat(docs, terms, present, freq, pos))) == r["rich_fnv"], r["q"]
    w.ix.close()


def test_phrases_inside_trees_against_the_reference(T, dev):
    """tests/golden/ref_phrase_trees.json — a multi-word phrase under an OR, inside a matchsome, under a NOT / an <optional>: 240 trees as the
    genuine reference compiled them, with ITS answers in AccumulatedScore mode (count, score sum, top-10) and in the default mode (matched
    terms and hits; DocumentsOnly crashes the reference on these shapes, SURVEY §0.10 — the oracle stands in for it there).  They run as
    TASK_TREE (k_tree.hpp): leaf bitmaps, the phrase leaves evaluated by hidden queries of the same batch."""
    g = json.load(open(os.path.join(GOLDEN, "ref_phrase_trees.json")))
    checked = hashed = 0
    for name, c in g["corpora"].items():
        w = World(T, dev, c["D"], c["V"], c["slots"], c["seed"])
        recs = [r for r in g["results"] if r["corpus"] == name]
        progs = [np.array(O.program_from_exec_tree(r["tree"]), dtype=np.uint32) for r in recs]
        sets, hashes, info = run_docs_only(w, progs)
        assert info["unsupported_queries"] == 0 and info["tree_queries"] > len(recs) // 2
        d, s, cnt, counts = run_scored(w, progs, 10)
        full = w.T.Batch(w.ix, progs, w.T.FLAG_ACCUMULATED_SCORE, topk=0)
        full.run()
        full.sync()
        for i, r in enumerate(recs):
            want, _ = w.ora.exec(progs[i], O.FLAG_DOCUMENTS_ONLY)
            assert np.array_equal(sets[i], want) and int(hashes[i]) == O.fnv1a_docs(want), r["q"]
            assert int(counts[i]) == r["n"], r["q"]
            top = r["top"]
            assert d[i, : len(top)].tolist() == [x[0] for x in top], r["q"]
            np.testing.assert_allclose(s[i, : len(top)], [x[1] for x in top], rtol=1e-5, err_msg=r["q"])
            assert np.array_equal(full.docset(i, r["n"]), sets[i]) or r["n"] != len(sets[i]), r["q"]
            assert abs(float(np.sum(full.scores(i, r["n"]))) - r["score_sum"]) <= 1e-5 * max(1.0, r["score_sum"]), r["q"]
            checked += 1
        full.close()
        for r, (docs, terms, present, freq, pos) in zip(recs, run_rich(w, progs)):
            assert len(docs) == r["rich_n"] and int(freq.sum()) == r["hits_total"], r["q"]
            assert int(sum(bin(int(x)).count("1") for x in present)) == r["terms_total"], r["q"]
            if r["rich_fnv"] is not None:  # (None: a shape whose default-mode positions the reference itself gets wrong — make_golden.py says which)
                assert str(O.fnv1a_u32_stream(rich_flat(docs, terms, present, freq, pos))) == r["rich_fnv"], r["q"]
                hashed += 1
        w.ix.close()
    assert checked == 240 and hashed >= 180


WIDE_TREES = ['t0 OR "t1 t2"', 't0 NOT ("t1 t2" t3)', "t0 OR (t1 t2) OR (t3 t4) OR (t5 t6) OR (t7 t8)", '[t0, "t1 t2", t3 t4, "t5 t6 t7"]', '"t0 t1" OR "t1 t2" OR "t2 t3"',
              "(t0 OR t1) (t2 OR t3) (t4 OR t5) (t6 OR t7) (t8 OR t9) (t10 OR t11) (t12 OR t13) (t14 OR t15) (t16 OR t17)", 't0 <"t1 t2">', '("t0 t1" OR t2) NOT "t3 t4"',
              "[t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11]", "t0 t1 t2 t3 t4 t5 t6 t7 t8 t9 t10 t11 t12 t13 t14 t15 t16 t17",
              't20 OR ((t0 OR "t1 t2") (t3 OR t4 OR t5) NOT (t6 "t7 t8"))']  # fmt: skip


@pytest.mark.parametrize("shape", [(2000, 200, 10, 42), (20000, 500, 12, 7), (300000, 3000, 10, 42)])
def test_trees_no_other_kernel_takes_match_oracle(T, dev, shape):
    """What used to be refused, against the oracle's iterator trees in all three modes: a multi-word phrase under an OR / NOT / matchsome /
    <optional>, several phrases in one tree, a general tree over more than eight distinct terms, a CNF of more than sixteen terms — also with
    masked documents.  What the planner still leaves out (status TRI_ERR_UNSUPPORTED, the rest of the batch runs): a tree of more than 64 nodes."""
    w = World(T, dev, *shape)
    big = " OR ".join(f"(t{2 * i} t{2 * i + 1})" for i in range(40))  # 40 conjunctions under an OR: 121 nodes
    texts = WIDE_TREES + ["t0 t1", big, '"t0 t1"']
    progs = [O.parse_query(t, some_min=2) for t in texts]
    status = [0] * len(WIDE_TREES) + [0, -3, 0]
    masked = np.array(sorted(set(np.random.default_rng(3).integers(1, shape[0], shape[0] // 7).tolist())), dtype=np.uint32)
    for mk in (None, masked):
        if mk is not None:
            w.ix.set_masked(mk)
            w.ora.set_masked(mk)
        b = T.Batch(w.ix, progs, T.FLAG_DOCUMENTS_ONLY, allow_unsupported=True)
        assert b.query_status().tolist() == status and b.info()["unsupported_queries"] == 1 and b.info()["tree_queries"] == len(WIDE_TREES)
        for rep in range(2):  # (a batch is re-run: the hidden queries' lists are rebuilt)
            b.run()
            b.sync()
            counts = b.counts()
            for i, t in enumerate(texts):
                want = w.ora.exec(progs[i], O.FLAG_DOCUMENTS_ONLY)[0] if not status[i] else np.zeros(0, np.uint32)
                assert int(counts[i]) == len(want), (t, rep)
                assert np.array_equal(b.docset(i, len(want)), want), (t, rep)
        b.close()
        ok = [i for i in range(len(texts)) if not status[i]]
        oprogs = [progs[i] for i in ok]
        for k in (10, 0):
            b = T.Batch(w.ix, oprogs, T.FLAG_ACCUMULATED_SCORE, topk=k)
            b.run()
            b.sync()
            counts = b.counts()
            tk = b.topk_results() if k else None
            for j, i in enumerate(ok):
                docs, scores = w.ora.exec(progs[i], O.FLAG_ACCUM_SCORE)
                assert int(counts[j]) == len(docs), texts[i]
                if k:
                    td, ts = w.ora.topk(docs, scores, k)
                    assert tk[0][j, : len(td)].tolist() == td.tolist(), texts[i]
                    np.testing.assert_allclose(tk[1][j, : len(td)], ts, rtol=1e-5, atol=0, err_msg=texts[i])
                else:
                    assert np.array_equal(b.docset(j, len(docs)), docs), texts[i]
                    np.testing.assert_allclose(b.scores(j, len(docs)), scores, rtol=1e-5, atol=0, err_msg=texts[i])
            b.close()
        rprogs = [progs[i] for i in ok if len(set(int(t) & 0x0FFFFFFF for t in progs[i] if int(t) >> 28 == T.OP_TERM)) <= 16]  # (the default mode reports at most 16 terms per query)
        for p, (docs, terms, present, freq, pos) in zip(rprogs, run_rich(w, rprogs)):
            wdocs, wflat, tt, ht = w.ora.exec_rich(p)
            assert np.array_equal(docs, wdocs)
            assert int(freq.sum()) == ht and int(sum(bin(int(x)).count("1") for x in present)) == tt
            assert np.array_equal(rich_flat(docs, terms, present, freq, pos), wflat)
    with pytest.raises(T.TrinityError):  # (a malformed program is still the caller's bug)
        T.Batch(w.ix, [np.array([T.tok(T.OP_AND, 2)], dtype=np.uint32)], T.FLAG_DOCUMENTS_ONLY)
    w.ix.set_masked(np.zeros(0, np.uint32))
    w.ix.close()


def test_large_batch_is_lowered_in_fragments(T, dev):
    """tri_batch_create lowers batches of 2048 queries and more on several host threads, each range of queries into a fragment of its own
    (term / phrase / scorer offsets relative to the fragment), joined in order: 6000 queries of every lowered kind — conjunctions, unions,
    CNFs, phrases (their DevPhrase rows and pterms cross fragment boundaries), NOT, general trees, and phrases under an OR (TASK_TREE: hidden queries, tree
    records and phrase rows cross fragment boundaries too) —
    give, query by query, what the same queries give in batches of 500 (one thread); a malformed program anywhere fails the batch."""
    w = World(T, dev, 3000, 300, 10, 43)
    rng = np.random.default_rng(5)
    shapes = ["t{0} t{1}", "t{0} OR t{1} OR t{2}", "t{0} (t{1} OR t{2})", '"t{0} t{1}"', '"t{0} t{1}" t{2}', "t{0} NOT t{1}", "t{0} OR (t{1} t{2})", 't{0} OR "t{1} t{2}"',
              "[t{0}, t{1}, t{2}]", "t{0} <t{1}>"]
    texts = [shapes[i % len(shapes)].format(*rng.choice(40, 3, replace=False)) for i in range(6000)]
    progs = [O.parse_query(t, some_min=2) for t in texts]
    for flags, topk in ((T.FLAG_DOCUMENTS_ONLY, 0), (T.FLAG_ACCUMULATED_SCORE, 10)):
        big = T.Batch(w.ix, progs, flags, topk=topk, allow_unsupported=True)
        big.run()
        big.sync()
        st, counts = big.query_status(), big.counts()
        tk = big.topk_results() if topk else None
        assert int((st != 0).sum()) == 0 == big.info()["unsupported_queries"] and big.info()["tree_queries"] == 600  # (the phrase under an OR, every 10th query: TASK_TREE)
        for lo in range(0, len(progs), 500):
            small = T.Batch(w.ix, progs[lo : lo + 500], flags, topk=topk, allow_unsupported=True)
            small.run()
            small.sync()
            assert small.query_status().tolist() == st[lo : lo + 500].tolist()
            assert small.counts().tolist() == counts[lo : lo + 500].tolist()
            if topk:
                d, s_, c = small.topk_results()
                assert np.array_equal(d, tk[0][lo : lo + 500]) and np.array_equal(s_, tk[1][lo : lo + 500]) and np.array_equal(c, tk[2][lo : lo + 500])
            small.close()
        for i in (0, 2999, 5999):  # and against the oracle, at the ends and in the middle
            if st[i] == 0:
                assert int(counts[i]) == len(w.ora.exec(progs[i], O.FLAG_DOCUMENTS_ONLY)[0]), texts[i]
        big.close()
    bad = list(progs)
    bad[4321] = np.array([T.tok(T.OP_AND, 2)], dtype=np.uint32)
    with pytest.raises(T.TrinityError):
        T.Batch(w.ix, bad, T.FLAG_DOCUMENTS_ONLY)
    w.ix.close()


# ------------------------------------------------------------------------------------------ hit payloads in the default mode
def test_hit_payloads_from_the_reference_segment(T, dev):
    """TRI_FLAG_MATCHED_TERMS | TRI_FLAG_HIT_PAYLOADS over the reference-written edge segment: for every term the fixture holds the
    reference's hash over each document's (freq, id) and each hit's (pos, payloadLen, the eight bytes of term_hit::payload) as
    Google::Decoder::materialize_hits leaves them (payload lengths changing from hit to hit, stale high bytes after a shorter payload).
    The single-term query `tK` reports exactly that stream; a two-term query reports both terms' payloads per match."""
    import base64

    g = json.load(open(os.path.join(GOLDEN, "ref_edge.json")))
    index = np.frombuffer(base64.b64decode(g["index_b64"]), dtype=np.uint8)
    terms = np.array(g["terms"], dtype=np.uint32)
    ix = T.Index(dev, index, terms, g["docsCnt"])
    ora = O.Index.wrap(index, terms, g["docsCnt"], g["postings"], g["sumTermHits"])
    try:
        recs = [r for r in g["results"] if r["cmd"] == "hits"]
        progs = [np.array([T.tok(T.OP_TERM, r["term"])], dtype=np.uint32) for r in recs] + [O.parse_query("t0 t1"), O.parse_query("t0 OR t4")]
        b = T.Batch(ix, progs, T.FLAG_MATCHED_TERMS | T.FLAG_HIT_PAYLOADS)
        b.run()
        b.sync()
        counts = b.counts()
        with_payload = 0
        for qi, r in enumerate(recs):
            n = int(counts[qi])
            docs = b.docset(qi, n)
            _, present, freq, pos = b.matched_terms(qi, n)
            lens, pl = b.matched_payloads(qi)
            assert len(lens) == len(pos) == int(freq.sum())
            h, at = 1469598103934665603, 0
            for i, d in enumerate(docs.tolist()):
                f = int(freq[i, 0])
                h = O.fnv1a_u32s([f, d], h)
                for k in range(at, at + f):
                    h = O.fnv1a_u32s([int(pos[k]), int(lens[k]), int(pl[k]) & 0xFFFFFFFF, int(pl[k]) >> 32], h)
                at += f
            assert n == r["docs"] and str(h) == r["fnv"], r["term"]
            with_payload += int(lens.any())
        assert with_payload >= 1
        # several terms per match: against the oracle (whose payload walk the CPU suite pins to the same fixture)
        for qi in (len(recs), len(recs) + 1):
            n = int(counts[qi])
            docs = b.docset(qi, n)
            qterms, present, freq, pos = b.matched_terms(qi, n)
            lens, pl = b.matched_payloads(qi)
            at = 0
            for i, d in enumerate(docs.tolist()):
                for k, t in enumerate(qterms.tolist()):
                    f = int(freq[i, k])
                    if not (int(present[i]) >> k) & 1:
                        assert f == 0
                        continue
                    it = O.PLI(ora, t)
                    assert it.advance(d) == d
                    wp, wl, ww = it.hits()
                    assert pos[at : at + f].tolist() == wp and lens[at : at + f].tolist() == wl and pl[at : at + f].tolist() == ww, (qi, d, t)
                    at += f
            assert at == len(pos)
        b.close()
        with pytest.raises(T.TrinityError):
            T.Batch(ix, progs[:1], T.FLAG_DOCUMENTS_ONLY | T.FLAG_HIT_PAYLOADS)
    finally:
        ix.close()


# ------------------------------------------------------------------------------------------ foreign / damaged chunks at upload
def test_upload_rejects_what_the_kernels_cannot_read(T, dev):
    """tri_index_upload validates the chunk format: a chunk whose non-final block holds fewer than 32 documents (legal to the reference's
    decoder, never written by its encoder; the kernels' tile and output layouts rely on full blocks) answers TRI_ERR_UNSUPPORTED, a
    truncated or corrupted chunk TRI_ERR_FORMAT — never a read past the buffer, never a wrong answer."""
    from trinity_amd import engine as E

    one = lambda docs: E.host_encode_google(docs, [1] * len(docs), list(range(1, len(docs) + 1)), [0, len(docs)])[0]
    a, b = one([1, 2]), one([3, 4, 5])  # two one-block chunks: [u16 0][block]
    chunk = np.concatenate([a, b[2:]])  # ... glued into one chunk of a 2-document and a 3-document block: documents 1, 2, 5, 6, 7
    with pytest.raises(T.TrinityError, match="rc=-3"):
        T.Index(dev, chunk, np.array([[5, 0, chunk.size]], dtype=np.uint32), 7)
    # the same bytes declared as what they are not
    with pytest.raises(T.TrinityError, match="rc=-4"):
        T.Index(dev, a, np.array([[3, 0, a.size]], dtype=np.uint32), 7)  # 2 documents in blocks, 3 declared
    good, terms = E.host_encode_google(np.arange(1, 101), [2] * 100, [1, 5] * 100, [0, 100])
    ix = T.Index(dev, good, terms, 100)
    ix.close()
    for cut in (1, 3, 17, good.size // 2):
        with pytest.raises(T.TrinityError, match="rc=-4"):
            T.Index(dev, good[: good.size - cut], np.array([[100, 0, good.size - cut]], dtype=np.uint32), 100)
    bad = good.copy()
    bad[3] = 0xF0  # the first block's length varint now claims five bytes
    with pytest.raises(T.TrinityError, match="rc=-4"):
        T.Index(dev, bad, terms, 100)
    with pytest.raises(T.TrinityError, match="rc=-4"):
        T.Index(dev, good, np.array([[100, 8, good.size]], dtype=np.uint32), 100)  # chunk outside the index


def test_lucene_upload_names_the_payload_it_reads(T, dev):
    """A LUCENE-coded segment whose ints() groups are neither PFOR128 nor FastPFor<4> words is refused at upload with TRI_ERR_FORMAT and
    a message that names both (never decoded into wrong postings); likewise a FastPFor-flavoured segment with a damaged group."""
    seg = T.Segment(3000, 100, 10, 42, codec=2)
    off, size = int(seg.terms[0, 1]), int(seg.terms[0, 2])
    assert int(seg.terms[0, 0]) >= 128
    bad = np.array(seg.index, copy=True)
    g = off + 14  # the first ints() group of term 0: [u8 L][L words]; word 0 = width | nexc << 8 | excwidth << 16
    assert bad[g] != 0
    bad[g + 1] = 33  # a packed width no PFOR128 group has (and the group's length no longer follows from its header word)
    with pytest.raises(T.TrinityError, match="rc=-4.*PFOR128.*FastPFor"):
        T.Index(dev, bad, seg.terms, seg.docs_cnt, codec=2, hits=seg.hits)
    segf = T.Segment(3000, 100, 10, 42, codec=3)
    off = int(segf.terms[0, 1])
    badf = np.array(segf.index, copy=True)
    g = off + 14
    assert badf[g] != 0 and badf[g + 1] == 128  # [u8 L][128 = the value count encodeArray stores first]...
    badf[g + 5] ^= 0x04  # ... [the offset of the metadata]: no longer 1 + 4 b
    with pytest.raises(T.TrinityError, match="rc=-4.*FastPFor"):
        T.Index(dev, badf, segf.terms, segf.docs_cnt, codec=2, hits=segf.hits)


def test_lucene_segment_with_the_reference_builds_payload_words(T, dev):
    """A LUCENE segment whose ints() groups carry FastPFor<4> words — the payload the reference's own lucene_codec build writes
    (lucene_codec.cpp:57-64; restated from the library's published algorithm, parity unpinned: csrc/fastpfor128.hpp) — loads (the upload
    transcribes every group to PFOR128, index and hits.data) and answers like the same corpus in the other encodings: decode of every
    term, docID sets incl. phrases (positions come from the transcribed hits.data), BM25 top-K, the default mode."""
    D, V = 30000, 1500
    w = World(T, dev, D, V, 10, 42, codec=2)
    segf = T.Segment(D, V, 10, 42, codec=3)
    assert segf.payload == "fastpfor" and segf.index.size != w.seg.index.size
    ixf = T.Index.from_segment(dev, segf)
    terms = np.arange(0, V, 7, dtype=np.uint32)
    df = segf.terms[terms, 0]
    docs, freqs, offs = ixf.decode_terms(terms, df)
    for i, t in enumerate(terms.tolist()[:80]):
        wd, wf = w.ora.decode_term(t)
        assert np.array_equal(docs[offs[i] : offs[i + 1]], wd) and np.array_equal(freqs[offs[i] : offs[i + 1]], wf), t
    texts = template_queries(w, 311, 10) + phrase_queries(w, 312, 6) + not_queries(w, 313, 3)
    progs = [O.parse_query(t) for t in texts]
    b = T.Batch(ixf, progs, T.FLAG_DOCUMENTS_ONLY)
    b.run()
    b.sync()
    counts = b.counts()
    for i, (t, p) in enumerate(zip(texts, progs)):
        want, _ = w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)
        assert np.array_equal(b.docset(i, int(counts[i])), want), t
    b.close()
    sb = T.Batch(ixf, progs, T.FLAG_ACCUMULATED_SCORE, topk=10)
    sb.run()
    sb.sync()
    d, s, c = sb.topk_results()
    for i, p in enumerate(progs):
        dd, ss = w.ora.exec(p, O.FLAG_ACCUM_SCORE)
        td, ts = w.ora.topk(dd, ss, 10)
        assert d[i, : len(td)].tolist() == td.tolist(), texts[i]
        np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0)
    sb.close()
    ixf.close()
    w.ix.close()


# ------------------------------------------------------------------------------------------ result gather behind the C-ABI (RCCL)
def test_gather_results_over_rccl_one_rank(T, dev, small):
    """tri_comm_* / tri_gather_results with a communicator of one rank (all this box has): what arrives in the [nranks][...] receive
    buffers is the batch's own device-resident result blocks.  (The torch.distributed form of the same gather — trinity_amd/dist.py —
    is what bench.py runs and what tests/test_dist_gloo.py covers at world_size 2.)"""
    import ctypes as C

    from trinity_amd import engine as E

    L = E.hip_lib()
    uid = (C.c_uint8 * 128)()
    E._check(L.tri_comm_unique_id(uid))
    comm = C.c_void_p()
    E._check(L.tri_comm_create(dev.h, uid, 0, 1, C.byref(comm)))
    w = small
    texts = template_queries(w, 141, 20)
    progs = [O.parse_query(t) for t in texts]
    b = T.Batch(w.ix, progs, T.FLAG_ACCUMULATED_SCORE, topk=10)
    b.run()
    nq, k = b.nq, 10
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    bufs = []
    for nbytes in (nq * 8, nq * k * 4, nq * k * 4, nq * 4):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), nbytes) == 0
        bufs.append((p, nbytes))
    E._check(L.tri_gather_results(b.h, comm, bufs[0][0], bufs[1][0], bufs[2][0], bufs[3][0]))
    b.sync()
    dev.sync()
    host = []
    for (p, nbytes), dt in zip(bufs, (np.uint64, np.uint32, np.float32, np.uint32)):
        a = np.zeros(nbytes // np.dtype(dt).itemsize, dtype=dt)
        assert hip.hipMemcpy(a.ctypes.data, p, nbytes, 2) == 0  # hipMemcpyDeviceToHost
        host.append(a)
        hip.hipFree(p)
    d, s, c = b.topk_results()
    assert np.array_equal(host[0], b.counts())
    assert np.array_equal(host[1].reshape(nq, k), d) and np.array_equal(host[2].reshape(nq, k), s) and np.array_equal(host[3], c)
    b.close()
    L.tri_comm_destroy(comm)


def test_gather_results_two_ranks_one_gpu(tmp_path):
    """tri_gather_results at world_size 2 on a one-GPU box: two processes, both on cuda:0, the communicator built over the test's own
    gloo transport (tri_comm_create_custom) — the blocks, their sizes and the [nranks][...] receive layout checked against the unsharded
    batch (tests/gather_worker.py)."""
    import subprocess
    import sys

    out = tmp_path / "gather.ok"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29633",
                        os.path.join(os.path.dirname(os.path.abspath(__file__)), "gather_worker.py"), str(out)], capture_output=True, text=True, timeout=600, env=env)  # fmt: skip
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert out.read_text().startswith("ok world=2")


def test_two_threads_compile_on_one_handle_while_a_third_runs(large):
    """include/trinity_hip.h, Threading (ABI 9): ONE device handle, two threads calling tri_batch_create side by side (the handle's two planner contexts — batches of 1500
    queries: large enough for the pools) while the main thread runs, awaits and reads back what they hand over — bench.py's loop.  Every batch answers what the same
    programs answer when compiled and run alone."""
    import queue
    import threading

    w, T = large, large.T
    sets = []
    for seed in (5, 6, 7):
        qs = T.gen_queries(w.V, seed, 1500, 2)
        progs = [and_prog(T, q) for q in qs.tolist()]
        b = T.Batch(w.ix, progs, T.FLAG_DOCUMENTS_ONLY)
        b.run()
        b.sync()
        sets.append((progs, b.counts().copy(), b.docset_hashes().copy()))
        b.close()
    ready, errors = queue.Queue(maxsize=2), []

    def compiler(i):
        try:
            for rep in range(12):
                k = (i + rep) % len(sets)
                ready.put((k, T.Batch(w.ix, sets[k][0], T.FLAG_DOCUMENTS_ONLY)))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))
        finally:
            ready.put(None)

    ths = [threading.Thread(target=compiler, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    done, seen, prev = 0, 0, None
    while done < 2:
        item = ready.get()
        if item is None:
            done += 1
            continue
        k, b = item
        b.run()  # behind the previous batch on the engine stream
        if prev is not None:
            pk, pb = prev
            pb.sync()
            assert np.array_equal(pb.counts(), sets[pk][1]) and np.array_equal(pb.docset_hashes(), sets[pk][2]), pk
            pb.close()
            seen += 1
        prev = (k, b)
    pk, pb = prev
    pb.sync()
    assert np.array_equal(pb.counts(), sets[pk][1]) and np.array_equal(pb.docset_hashes(), sets[pk][2])
    pb.close()
    for t in ths:
        t.join()
    assert not errors and seen + 1 == 24, (errors, seen)


def test_two_host_threads_two_device_handles(T):
    """SURVEY §8(b) threading: the ABI is called concurrently from two host threads, each with its own tri_dev (own stream) on the same
    GPU, own index upload and own batches — exec_query's re-entrancy per thread (exec.cpp:12).  Every thread's results equal the ones
    computed alone beforehand, run after run."""
    import threading

    D, V = 20000, 2000
    seg = T.Segment(D, V, 10, 42)
    ora = O.Index.wrap(seg.index, seg.terms, seg.docs_cnt, seg.sum_terms_docs, seg.sum_term_hits)
    texts = [["t0 t1", "t2 OR t3 OR t4", "t0 t1 (t2 OR t3)", '"t0 t1"', "t5 NOT t1"], ["t1 t2", "t0 OR t9", "t3 t4 t5", '"t1 t2" t0', "[t0, t1, t2]"]]
    progs = [[O.parse_query(t, some_min=2) for t in tt] for tt in texts]
    want = [[ora.exec(p, O.FLAG_ACCUM_SCORE) for p in pp] for pp in progs]
    errors = []

    def worker(i):
        try:
            dev = T.Device(0)
            ix = T.Index.from_segment(dev, seg)
            for rep in range(6):
                bd = T.Batch(ix, progs[i], T.FLAG_DOCUMENTS_ONLY)
                bs = T.Batch(ix, progs[i], T.FLAG_ACCUMULATED_SCORE, topk=10)
                bd.run()
                bs.run()
                bd.sync()
                bs.sync()
                d, s, c = bs.topk_results()
                for q, (docs, scores) in enumerate(want[i]):
                    assert np.array_equal(bd.docset(q), docs), (i, rep, texts[i][q])
                    td, ts = ora.topk(docs, scores, 10)
                    assert d[q, : len(td)].tolist() == td.tolist(), (i, rep, texts[i][q])
                    np.testing.assert_allclose(s[q, : len(td)], ts, rtol=1e-5, atol=0)
                bd.close()
                bs.close()
            ix.close()
            dev.close()
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors


def test_device_encoder_refuses_what_the_reference_encoder_would_not_take(T, dev):
    """tri_encode_google validates on the host: positions non-descending and non-zero within a posting (google_codec.cpp:42-49), freqs[]
    within positions[] — TRI_ERR_INVALID instead of bytes the reference's encoder would never write."""
    docs, freqs, tf = np.array([1, 5], np.uint32), np.array([2, 1], np.uint32), np.array([0, 2], np.uint64)
    dev.encode_google(docs, freqs, np.array([3, 7, 2], np.uint16), tf)  # fine
    for bad in (np.array([7, 3, 2], np.uint16), np.array([0, 3, 2], np.uint16), np.array([3, 7], np.uint16)):
        with pytest.raises(T.TrinityError):
            dev.encode_google(docs, freqs, bad, tf)


# ------------------------------------------------------------------------------------------ write side on the device
def random_postings(rng, nterms):
    """Postings that reach every corner of the encoder: empty terms, 1 / 31 / 32 / 33 / 64 / 65 documents, runs long enough for skiplist
    entries, deltas of every varint length, frequencies 0 .. 300, positions up to 65535 with repeats."""
    docs, freqs, pos, tf = [], [], [], [0]
    sizes = [0, 1, 31, 32, 33, 64, 65, 300, 1000, 2, 5]
    for t in range(nterms):
        n = sizes[t % len(sizes)] if t < 3 * len(sizes) else int(rng.integers(0, 200))
        scale = [1, 3, 200, 20000, 3_000_000][t % 5]
        d = np.cumsum(rng.integers(1, scale + 1, size=n, dtype=np.int64))
        d = d[d < 2**32 - 1]
        f = np.where(rng.random(d.size) < 0.1, 0, rng.integers(1, 4, size=d.size))
        if d.size and t % 7 == 0:
            f[int(rng.integers(0, d.size))] = 300
        for k in f.tolist():
            p = np.sort(rng.integers(1, [12, 200, 65536][t % 3], size=k))
            pos += p.tolist()
        docs += d.tolist()
        freqs += f.tolist()
        tf.append(len(docs))
    return np.array(docs, dtype=np.uint32), np.array(freqs, dtype=np.uint32), np.array(pos, dtype=np.uint16), np.array(tf, dtype=np.uint64)


def test_google_encoder_on_the_device(T, dev):
    """tri_encode_google against the host encoder (byte-identical to the reference's Codecs::Google::Encoder, tests/test_abi.py and the
    oracle's index FNV): the same postings give the same `index` bytes and term table — skiplist entries across terms included —,
    and the encoded segment decodes back to the postings on the device."""
    from trinity_amd import engine as E

    rng = np.random.default_rng(5)
    for nterms in (1, 40, 400, 3000):  # (3000 terms: a few hundred thousand postings — the scans run over several chunks, k_encode.hpp)
        docs, freqs, pos, tf = random_postings(rng, nterms)
        if nterms == 3000:
            assert docs.size > 4 * 32768
        got, gterms = dev.encode_google(docs, freqs, pos, tf)
        want, wterms = E.host_encode_google(docs, freqs, pos, tf)
        assert np.array_equal(gterms, wterms), nterms
        assert got.size == want.size and np.array_equal(got, want), (nterms, int(np.argmax(got[: want.size] != want[: got.size])))
        if nterms == 400:  # round trip through the read side
            ix = T.Index(dev, got, gterms, int(docs.max()))
            df = gterms[:, 0].astype(np.int64)
            d2, f2, offs = ix.decode_terms(np.arange(nterms, dtype=np.uint32), df)
            assert np.array_equal(d2, docs) and np.array_equal(f2, freqs)
            ix.close()


@pytest.mark.parametrize("payloads", [False, True])
def test_commit_on_the_device(T, dev, payloads):
    """tri_commit_google — SegmentIndexSession::commit (indexer.cpp:311-478): a session's postings in INSERTION order (document after document, a
    document's terms in any order) are sorted on the device by (termID & 31, termID, documentID) — the order the reference's commit walks its 32 buckets
    in —, gathered and encoded there.  The bytes, the committed terms' order and their term table equal the host encoder's over the same postings handed
    over term after term in that order; the field statistics are the session's; what commit would refuse is refused."""
    from trinity_amd import engine as E

    def slices(hit0, idx):  # the hits of the postings idx, one posting after the other
        n = (hit0[idx + 1] - hit0[idx]).astype(np.int64)
        return np.repeat(hit0[idx] - np.concatenate([[0], np.cumsum(n)[:-1]]), n) + np.arange(int(n.sum())) if idx.size else np.zeros(0, np.int64)

    rng = np.random.default_rng(17)
    for nterms in (1, 60, 2500):
        docs, freqs, pos, tf = random_postings(rng, nterms)
        hit0 = np.concatenate([[0], np.cumsum(freqs.astype(np.int64))])
        plen = rng.integers(0, 9, size=pos.size).astype(np.uint8) if payloads else None
        pval = rng.integers(0, 2**63, size=pos.size, dtype=np.uint64) if payloads else None
        if payloads:
            pval &= (np.uint64(1) << (plen.astype(np.uint64) * np.uint64(8))) - np.uint64(1)  # (only the payload's own bytes count)
            pval[plen == 8] = rng.integers(0, 2**63, size=int((plen == 8).sum()), dtype=np.uint64)
        tids = rng.choice(np.arange(1, 50 * nterms + 64, dtype=np.uint32), size=nterms, replace=False)  # (many share their low five bits)
        term_of = np.repeat(np.arange(nterms), np.diff(tf).astype(np.int64))
        # the session: documents in a random order, every document's postings together, its terms in a random order
        uniq, inv = np.unique(docs, return_inverse=True)
        order = np.lexsort((rng.random(docs.size), rng.permutation(uniq.size)[inv]))  # (grouped by document, the documents in a random order)
        s_terms, s_docs, s_freqs = tids[term_of[order]], docs[order], freqs[order]
        take = slices(hit0, order)
        s_pos = pos[take]
        got, gtids, gterms, stats = dev.commit_google(s_terms, s_docs, s_freqs, s_pos, None if plen is None else plen[take], None if pval is None else pval[take])
        # what commit feeds the encoder: the terms that have postings, by (termID & 31, termID); each one's documents ascending
        live = [t for t in range(nterms) if tf[t + 1] > tf[t]]
        live.sort(key=lambda t: (int(tids[t]) & 31, int(tids[t])))
        sel = np.concatenate([np.arange(tf[t], tf[t + 1], dtype=np.int64) for t in live]) if live else np.zeros(0, np.int64)
        wtf = np.concatenate([[0], np.cumsum([int(tf[t + 1] - tf[t]) for t in live])]).astype(np.uint64)
        wtake = slices(hit0, sel)
        want, wterms = E.host_encode_google(docs[sel], freqs[sel], pos[wtake], wtf, None if plen is None else plen[wtake], None if pval is None else pval[wtake])
        assert gtids.tolist() == [int(tids[t]) for t in live], nterms
        assert np.array_equal(gterms, wterms), nterms
        assert got.size == want.size and np.array_equal(got, want), (nterms, int(np.argmax(got[: want.size] != want[: got.size])))
        assert stats == {"docs_cnt": len(set(docs.tolist())), "sum_terms_docs": int(docs.size), "sum_term_hits": int(freqs.sum()), "total_terms": len(live)}
        if not payloads:  # the same session committed through the Lucene-shaped codec's encoder (commit is codec-agnostic: indexer.cpp:323)
            from trinity_amd import hostplan as HP

            li, lh, ltids, lterms, lstats = dev.commit_lucene(s_terms, s_docs, s_freqs, s_pos)
            wi, wh, wt = HP.lucene_encode(docs[sel], freqs[sel], pos[wtake], wtf)
            assert ltids.tolist() == gtids.tolist() and np.array_equal(lterms, wt) and lstats == stats, nterms
            assert np.array_equal(li, wi) and np.array_equal(lh, wh), nterms
    # refused: the same (term, document) twice; document 0; positions out of order
    with pytest.raises(T.TrinityError, match="twice"):
        dev.commit_google([7, 7], [5, 5], [1, 1], [3, 4])
    with pytest.raises(T.TrinityError, match="document 0"):
        dev.commit_google([7], [0], [1], [3])
    with pytest.raises(T.TrinityError, match="positions"):
        dev.commit_google([7], [5], [2], [9, 3])
    got, gtids, gterms, stats = dev.commit_google(np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint16))
    assert got.size == 0 and gtids.size == 0 and stats["total_terms"] == 0


def test_merge_on_the_device(T, dev):
    """tri_merge_google — Codecs::Google::IndexSession::merge (google_codec.cpp:186-438) over a whole dictionary: three segments with overlapping
    dictionaries and documentIDs, each masked by its own set; per output term the union of the documents, a document from the MOST RECENT participant that
    holds it, dropped when that participant's masked set holds it, hits and payloads carried over.  The merged `index` bytes and term table equal the host
    encoder's over the postings a plain restatement of the reference's k-way walk keeps; a term that keeps nothing stays as an empty chunk."""
    from trinity_amd import engine as E

    rng = np.random.default_rng(23)
    G, D, nparts = 260, 6000, 3
    segs = []
    for p in range(nparts):
        held = np.sort(rng.choice(G, size=int(G * 0.7), replace=False))
        docs, freqs, pos, plen, pval, tf = [], [], [], [], [], [0]
        for g in held.tolist():
            n = [1, 31, 32, 33, 70, 400][g % 6] if g % 11 else 0  # (some terms hold no document in this segment)
            d = np.sort(rng.choice(np.arange(1, D), size=n, replace=False))
            f = rng.integers(0, 4, size=n)
            for k in f.tolist():
                pp = np.sort(rng.integers(1, 2000, size=k))
                pl = rng.integers(0, 9, size=k) if g % 3 == 0 else np.zeros(k, np.int64)
                pos += pp.tolist()
                plen += pl.tolist()
                pval += [int(rng.integers(0, 2**62)) & ((1 << (8 * int(x))) - 1) for x in pl.tolist()]
            docs += d.tolist()
            freqs += f.tolist()
            tf.append(len(docs))
        a = dict(held=held, docs=np.array(docs, np.uint32), freqs=np.array(freqs, np.uint32), pos=np.array(pos, np.uint16), plen=np.array(plen, np.uint8),
                 pval=np.array(pval, np.uint64), tf=np.array(tf, np.uint64), masked=np.sort(rng.choice(np.arange(1, D), size=D // 6, replace=False)).astype(np.uint32))  # fmt: skip
        a["hit0"] = np.concatenate([[0], np.cumsum(a["freqs"].astype(np.int64))])
        index, terms = E.host_encode_google(a["docs"], a["freqs"], a["pos"], a["tf"], a["plen"], a["pval"])
        a["ix"] = T.Index(dev, index, terms, D)
        a["ix"].set_masked(a["masked"])
        segs.append(a)
    out_terms = [g for g in range(G) if any(g in s["held"] for s in segs)]
    part_terms = np.full((len(out_terms), nparts), 0xFFFFFFFF, dtype=np.uint32)
    for t, g in enumerate(out_terms):
        for p, s in enumerate(segs):
            k = np.searchsorted(s["held"], g)
            if k < len(s["held"]) and s["held"][k] == g:
                part_terms[t, p] = k
    got, gterms, stats = dev.merge_google([s["ix"] for s in segs], part_terms)
    # the reference's walk, restated: lowest documentID first; among the participants that hold it the most recent (lowest index) supplies it; its own masked set decides
    docs, freqs, pos, plen, pval, tf = [], [], [], [], [], [0]
    for t, g in enumerate(out_terms):
        best = {}
        for p, s in enumerate(segs):
            k = int(part_terms[t, p])
            if k == 0xFFFFFFFF:
                continue
            for i in range(int(s["tf"][k]), int(s["tf"][k + 1])):
                best.setdefault(int(s["docs"][i]), (p, i))
        for dd in sorted(best):
            p, i = best[dd]
            s = segs[p]
            km = int(np.searchsorted(s["masked"], dd))
            if km < len(s["masked"]) and s["masked"][km] == dd:
                continue
            h0, h1 = int(s["hit0"][i]), int(s["hit0"][i + 1])
            docs.append(dd)
            freqs.append(int(s["freqs"][i]))
            pos += s["pos"][h0:h1].tolist()
            plen += s["plen"][h0:h1].tolist()
            pval += s["pval"][h0:h1].tolist()
        tf.append(len(docs))
    want, wterms = E.host_encode_google(np.array(docs, np.uint32), np.array(freqs, np.uint32), np.array(pos, np.uint16), np.array(tf, np.uint64), np.array(plen, np.uint8),
                                        np.array(pval, np.uint64))  # fmt: skip
    assert np.array_equal(gterms, wterms)
    assert got.size == want.size and np.array_equal(got, want), int(np.argmax(got[: want.size] != want[: got.size]))
    assert int((gterms[:, 0] == 0).sum()) > 0  # (terms that keep nothing are there, empty)
    assert stats["sum_terms_docs"] == len(docs) and stats["sum_term_hits"] == int(np.sum(freqs)) and stats["total_terms"] == int((wterms[:, 0] > 0).sum())
    # the merged segment reads back: every term's postings through the decoder
    ix = T.Index(dev, got, gterms, D)
    d2, f2, offs = ix.decode_terms(np.arange(len(out_terms), dtype=np.uint32), gterms[:, 0].astype(np.int64))
    assert np.array_equal(d2, np.array(docs, np.uint32)) and np.array_equal(f2, np.array(freqs, np.uint32))
    ix.close()
    # one participant, nothing masked: the postings come through unchanged
    segs[0]["ix"].set_masked(np.zeros(0, np.uint32))
    one, oterms, _ = dev.merge_google([segs[0]["ix"]], np.arange(len(segs[0]["held"]), dtype=np.uint32).reshape(-1, 1))
    w1, t1 = E.host_encode_google(segs[0]["docs"], segs[0]["freqs"], segs[0]["pos"], segs[0]["tf"], segs[0]["plen"], segs[0]["pval"])
    assert np.array_equal(one, w1) and np.array_equal(oterms, t1)
    for s in segs:
        s["ix"].close()


def test_lucene_merge_on_the_device(T, dev):
    """tri_merge_lucene — Codecs::Lucene::IndexSession::merge (lucene_codec.cpp:963-1396) over a whole dictionary: three Lucene-shaped segments (uploaded with
    their hits.data) with overlapping dictionaries and documentIDs, each masked by its own set; per output term the union of the documents, a document from the
    MOST RECENT participant that holds it, dropped when that participant's masked set holds it, positions carried over.  `index`, `hits.data` and the term table
    equal the host encoder's over the postings the restated walk keeps (the walk tests/golden/ref_merge.json pins for the Google codec), lists across the
    128-document block cadence and the 128-hit block cadence included; the merged segment reads back through upload and decode."""
    from trinity_amd import hostplan as HP

    rng = np.random.default_rng(29)
    G, D, nparts = 200, 9000, 3
    segs = []
    for p in range(nparts):
        held = np.sort(rng.choice(G, size=int(G * 0.7), replace=False))
        docs, freqs, pos, tf = [], [], [], [0]
        for g in held.tolist():
            n = [1, 31, 127, 128, 129, 300, 700][g % 7] if g % 11 else 0  # (some terms hold no document in this segment)
            d = np.sort(rng.choice(np.arange(1, D), size=n, replace=False))
            f = rng.integers(0, 5, size=n)
            for k in f.tolist():
                pos += np.sort(rng.choice(np.arange(1, 3000), size=k, replace=False)).tolist()
            docs += d.tolist()
            freqs += f.tolist()
            tf.append(len(docs))
        a = dict(held=held, docs=np.array(docs, np.uint32), freqs=np.array(freqs, np.uint32), pos=np.array(pos, np.uint16), tf=np.array(tf, np.uint64),
                 masked=np.sort(rng.choice(np.arange(1, D), size=D // 6, replace=False)).astype(np.uint32))  # fmt: skip
        a["hit0"] = np.concatenate([[0], np.cumsum(a["freqs"].astype(np.int64))])
        index, hits, terms = HP.lucene_encode(a["docs"], a["freqs"], a["pos"], a["tf"])
        a["ix"] = T.Index(dev, index, terms, D, codec=2, hits=hits)
        a["ix"].set_masked(a["masked"])
        segs.append(a)
    out_terms = [g for g in range(G) if any(g in s["held"] for s in segs)]
    part_terms = np.full((len(out_terms), nparts), 0xFFFFFFFF, dtype=np.uint32)
    for t, g in enumerate(out_terms):
        for p, s in enumerate(segs):
            k = np.searchsorted(s["held"], g)
            if k < len(s["held"]) and s["held"][k] == g:
                part_terms[t, p] = k
    gi, gh, gterms, stats = dev.merge_lucene([s["ix"] for s in segs], part_terms)
    docs, freqs, pos, tf = [], [], [], [0]
    for t, g in enumerate(out_terms):
        best = {}
        for p, s in enumerate(segs):
            k = int(part_terms[t, p])
            if k == 0xFFFFFFFF:
                continue
            for i in range(int(s["tf"][k]), int(s["tf"][k + 1])):
                best.setdefault(int(s["docs"][i]), (p, i))
        for dd in sorted(best):
            p, i = best[dd]
            s = segs[p]
            km = int(np.searchsorted(s["masked"], dd))
            if km < len(s["masked"]) and s["masked"][km] == dd:
                continue
            h0, h1 = int(s["hit0"][i]), int(s["hit0"][i + 1])
            docs.append(dd)
            freqs.append(int(s["freqs"][i]))
            pos += s["pos"][h0:h1].tolist()
        tf.append(len(docs))
    wi, wh, wterms = HP.lucene_encode(np.array(docs, np.uint32), np.array(freqs, np.uint32), np.array(pos, np.uint16), np.array(tf, np.uint64))
    assert np.array_equal(gterms, wterms)
    assert gi.size == wi.size and np.array_equal(gi, wi), ("index", int(np.argmax(gi[: wi.size] != wi[: gi.size])))
    assert gh.size == wh.size and np.array_equal(gh, wh), ("hits.data", int(np.argmax(gh[: wh.size] != wh[: gh.size])))
    assert int((gterms[:, 0] == 0).sum()) > 0  # (terms that keep nothing are there, empty)
    assert stats["sum_terms_docs"] == len(docs) and stats["sum_term_hits"] == int(np.sum(freqs)) and stats["total_terms"] == int((wterms[:, 0] > 0).sum())
    ix = T.Index(dev, gi, gterms, D, codec=2, hits=gh)
    d2, f2, offs = ix.decode_terms(np.arange(len(out_terms), dtype=np.uint32), gterms[:, 0].astype(np.int64))
    assert np.array_equal(d2, np.array(docs, np.uint32)) and np.array_equal(f2, np.array(freqs, np.uint32))
    # ... and a phrase over the merged segment's positions equals the same phrase evaluated per winning posting (the hits came through in order)
    ix.close()
    # one participant, nothing masked: the postings come through unchanged
    s0 = segs[0]
    s0["ix"].set_masked(np.zeros(0, np.uint32))
    only = np.arange(len(s0["held"]), dtype=np.uint32).reshape(-1, 1)
    i1, h1_, t1, _ = dev.merge_lucene([s0["ix"]], only)
    wi1, wh1, wt1 = HP.lucene_encode(s0["docs"], s0["freqs"], s0["pos"], s0["tf"])
    assert np.array_equal(i1, wi1) and np.array_equal(h1_, wh1) and np.array_equal(t1, wt1)
    # a Google-coded participant, or one uploaded without its hits.data, is refused
    with pytest.raises(T.TrinityError):
        dev.merge_lucene([T.Index(dev, wi1, wt1, D, codec=2)], only)
    for s in segs:
        s["ix"].close()


def test_lucene_encoder_on_the_device(T, dev):
    """tri_encode_lucene — Codecs::Lucene::Encoder (lucene_codec.cpp:163-388) with the PFOR128 payload, one unit of csrc/lucene_enc_units.hpp per lane:
    `index`, `hits.data` and the term table equal the sequential host encoder's (lucene_encoder.hpp: the writer of every Lucene-shaped segment the engine
    reads) byte for byte, and the encoded segment reads back through the upload and the decoder."""
    from test_fastpfor import lucene_postings

    from trinity_amd import hostplan as HP

    rng = np.random.default_rng(3)
    for nterms in (1, 36, 300, 1500):
        docs, freqs, pos, tf = lucene_postings(rng, nterms)
        gi, gh, gt = dev.encode_lucene(docs, freqs, pos, tf)
        wi, wh, wt = HP.lucene_encode(docs, freqs, pos, tf)
        assert np.array_equal(gt, wt), nterms
        assert gi.size == wi.size and np.array_equal(gi, wi), (nterms, "index", int(np.argmax(gi[: wi.size] != wi[: gi.size])))
        assert gh.size == wh.size and np.array_equal(gh, wh), (nterms, "hits.data", int(np.argmax(gh[: wh.size] != wh[: gh.size])))
        if nterms == 300:  # round trip through the read side
            ix = T.Index(dev, gi, gt, int(docs.max()), codec=2, hits=gh)
            df = gt[:, 0].astype(np.int64)
            d2, f2, offs = ix.decode_terms(np.arange(nterms, dtype=np.uint32), df)
            assert np.array_equal(d2, docs) and np.array_equal(f2, freqs)
            ix.close()
    with pytest.raises(T.TrinityError):
        dev.encode_lucene([5, 3], [0, 0], [], [0, 2])  # (documents must ascend within a term)


def test_google_encoder_on_the_device_with_payloads(T, dev):
    """tri_encode_google_payloads against the host encoder (whose payload bytes the reference-written edge segment pins, tests/test_abi.py):
    hits with payloads of 0 .. 8 bytes whose length changes from hit to hit or stays (both arms of the flag bit, google_codec.cpp:59-66),
    the length state restarting with every document, a counted position-0 hit with a payload — byte-identical; and the encoded segment,
    uploaded, hands the same payloads back in the default mode (tri_batch_matched_payloads)."""
    from trinity_amd import engine as E

    rng = np.random.default_rng(9)
    for nterms in (1, 37, 200):
        docs, freqs, pos, tf = random_postings(rng, nterms)
        style = rng.integers(0, 3, size=pos.size)  # per hit: no payload / the previous one's length again / a fresh length
        plen = np.where(style == 0, 0, rng.integers(1, 9, size=pos.size)).astype(np.uint8)
        same = np.flatnonzero(style == 1)
        plen[same[same > 0]] = plen[same[same > 0] - 1]
        pv = rng.integers(0, 2**63, size=pos.size, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=pos.size, dtype=np.uint64)
        # a counted position-0 hit: the first hit of some documents, with a payload
        first = np.cumsum(np.concatenate([[0], freqs[:-1]])).astype(np.int64)
        for h in first[(freqs > 0) & (rng.random(freqs.size) < 0.05)].tolist():
            pos[h], plen[h] = 0, max(1, int(plen[h]))
        want, wterms = E.host_encode_google(docs, freqs, pos, tf, plen, pv)
        got, gterms = dev.encode_google(docs, freqs, pos, tf, plen, pv)
        assert np.array_equal(gterms, wterms), nterms
        assert got.size == want.size and np.array_equal(got, want), nterms
    for bad_plen, bad_pos in ((9, 5), (0, 0)):  # a payload of nine bytes; a payload-less hit at position 0
        with pytest.raises(T.TrinityError):
            dev.encode_google(np.array([1], np.uint32), np.array([1], np.uint32), np.array([bad_pos], np.uint16), np.array([0, 1], np.uint64),
                              np.array([bad_plen], np.uint8), np.array([7], np.uint64))
    # round trip through the engine: the payloads of a term's hits as the default mode reports them
    docs, freqs, tf = np.array([3, 9, 10], np.uint32), np.array([2, 1, 3], np.uint32), np.array([0, 3], np.uint64)
    pos = np.array([1, 4, 2, 5, 6, 9], np.uint16)
    plen = np.array([2, 2, 0, 8, 0, 3], np.uint8)
    pv = np.array([0x1122, 0x3344, 0, 0x8877665544332211, 0, 0xABCDEF], np.uint64)
    index, terms = dev.encode_google(docs, freqs, pos, tf, plen, pv)
    ix = T.Index(dev, index, terms, 10)
    b = T.Batch(ix, [O.parse_query("t0")], T.FLAG_MATCHED_TERMS | T.FLAG_HIT_PAYLOADS)
    b.run()
    b.sync()
    _, _, _, positions = b.matched_terms(0, 3)
    lens, words = b.matched_payloads(0)
    assert positions.tolist() == pos.tolist() and lens.tolist() == plen.tolist()
    # (term_hit::payload keeps the bytes a shorter payload does not overwrite — google_codec.cpp:533-594; a document starts from zero)
    assert [int(x) & ((1 << (8 * int(l))) - 1) for x, l in zip(words.tolist(), plen.tolist())] == [int(x) & ((1 << (8 * int(l))) - 1) for x, l in zip(pv.tolist(), plen.tolist())]
    b.close()
    ix.close()


def test_device_encoder_reproduces_the_synthetic_segment(T, dev):
    """The tiny corpus' postings (read back through the oracle: documents, frequencies, positions) re-encoded on the device give the
    segment's own bytes — the bytes the reference's encoder writes for this corpus (tests/test_oracle.py pins their FNV)."""
    seg = T.Segment(2000, 200, 10, 42)
    ora = O.Index.wrap(seg.index, seg.terms, seg.docs_cnt, seg.sum_terms_docs, seg.sum_term_hits)
    docs, freqs, pos, tf, kept = [], [], [], [0], []
    for t in range(200):
        if not int(seg.terms[t, 0]):
            continue  # (the segment writes nothing for a term without documents)
        it = O.PLI(ora, t)
        while True:
            d = it.next()
            if d == O.DOCIDS_END:
                break
            docs.append(d)
            freqs.append(it.freq())
            pos += it.positions()
        tf.append(len(docs))
        kept.append(t)
    got, gterms = dev.encode_google(docs, freqs, pos, tf)
    assert np.array_equal(got, np.asarray(seg.index))
    assert np.array_equal(gterms, np.asarray(seg.terms)[kept])


# ------------------------------------------------------------------------------------------ collections of segments
def test_collection_of_two_segments(T, dev):
    """IndexSourcesCollection semantics (index_source.cpp:3-30): an older segment whose documents 1..6000 were re-indexed into a newer
    one — the older source is masked by the newer one's documents, the same queries run over both, and the application sees the union:
    docID sets source after source, match counts added up, ONE top-K over both (merged on the device from the parts' partial lists)."""
    old = World(T, dev, 20000, 2000, 10, 42)
    new = World(T, dev, 6000, 2000, 10, 7)
    try:
        old.ix.set_masked(np.arange(1, 6001, dtype=np.uint32))
        texts = template_queries(old, 131, 20) + not_queries(old, 132, 4) + ["t0 t1", "t0 OR t1 OR t2 OR t3", "t5", "[t0, t1, t2]"]
        progs = [O.parse_query(t, some_min=2) for t in texts]
        for opts in ({}, {"dense_min_postings": 0}):
            with options(dev, **opts):
                parts = [T.Batch(w.ix, progs, T.FLAG_DOCUMENTS_ONLY) for w in (old, new)]
                cb = T.CollectionBatch(parts)
                cb.run()
                cb.sync()
                counts = cb.counts()
                sparts = [T.Batch(w.ix, progs, T.FLAG_ACCUMULATED_SCORE, topk=10) for w in (old, new)]
                sb = T.CollectionBatch(sparts)
                sb.run()
                sb.sync()
                d, s, c = sb.topk_results()
                scounts = sb.counts()
            for i, (t, p) in enumerate(zip(texts, progs)):
                do, so = old.ora.exec(p, O.FLAG_ACCUM_SCORE)
                keep = do > 6000
                dn, sn = new.ora.exec(p, O.FLAG_ACCUM_SCORE)
                want = np.concatenate([do[keep], dn])
                assert int(counts[i]) == len(want) == int(scounts[i]), (opts, t)
                assert np.array_equal(cb.docset(i, len(want)), want), (opts, t)
                td, ts = old.ora.topk(want, np.concatenate([so[keep], sn]), 10)
                assert int(c[i]) == len(td) and d[i, : len(td)].tolist() == td.tolist(), (opts, t)
                np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0)
            cb.close()
            sb.close()
            for b in parts + sparts:
                b.close()
    finally:
        old.ix.close()
        new.ix.close()


# ------------------------------------------------------------------------------------------ masked documents
@pytest.mark.parametrize("codec", [1, 2])
def test_masked_documents_are_dropped(T, dev, codec):
    """masked_documents_registry::test (docidupdates.h:90-119; exec.cpp:914-975): a document masked by a newer segment never
    reaches consider().  Every query shape, both matching kernels, scored and not; then the set is cleared again."""
    w = World(T, dev, 30000, 3000, 10, 42, codec=codec)
    rng = np.random.default_rng(5)
    masked = np.unique(np.concatenate([rng.integers(1, 30001, 4000), np.arange(100, 400), [1, 30000, 29999]])).astype(np.uint32)
    texts = (template_queries(w, 81, 6) + not_queries(w, 82, 4) + phrase_queries(w, 83, 3) + ["t0", "t7", "t0 t1", "t0 OR t1", "t2500", "t0 NOT t1"])
    progs = [O.parse_query(t) for t in texts]
    base_sets, _, _ = run_docs_only(w, progs)
    w.ix.set_masked(masked)
    w.ora.set_masked(masked)
    try:
        for dense_min in (None, 0):  # planner's choice, then the bitmap-window kernel forced
            with options(dev, **({} if dense_min is None else {"dense_min_postings": dense_min})):
                sets, hashes, _ = run_docs_only(w, progs)
            dropped = 0
            for t, p, got, h, full in zip(texts, progs, sets, hashes, base_sets):
                want, _ = w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)
                assert np.array_equal(got, want), (t, len(got), len(want))
                assert np.array_equal(got, full[~np.isin(full, masked)]), t  # == the unmasked result minus the masked set
                assert int(h) == O.fnv1a_docs(want)
                dropped += len(full) - len(got)
            assert dropped > 1000
        scored = [p for t, p in zip(texts, progs)]
        for opts in ({}, {"dense_min_postings": 0}, {"dense_min_postings": 0, "fused": 0}):  # planner's choice; one-pass scored windows forced; match-then-score forced
            with options(dev, **opts):
                d, s, c, counts = run_scored(w, scored, 20)
            for i, p in enumerate(scored):
                docs, scores = w.ora.exec(p, O.FLAG_ACCUM_SCORE)
                assert int(counts[i]) == len(docs), (opts, texts[i])
                td, ts = w.ora.topk(docs, scores, 20)
                assert d[i, : len(td)].tolist() == td.tolist(), (opts, texts[i])
                np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0)
    finally:
        w.ix.set_masked(np.zeros(0, np.uint32))
        w.ora.set_masked(np.zeros(0, np.uint32))
    sets, _, _ = run_docs_only(w, progs)
    for got, full in zip(sets, base_sets):
        assert np.array_equal(got, full)
    w.ix.close()


@pytest.mark.parametrize("sim", ["tfidf", "trivial"])
@pytest.mark.parametrize("world", ["small", "small_l"])
def test_other_similarities_match_oracle(request, world, sim):
    """IndexSourcesCollectionTFIDFScorer / ...TrivialScorer (similarity.h:75-163 / 56-72) through k_score and k_phrase."""
    w = request.getfixturevalue(world)
    texts = template_queries(w, 91, 6) + phrase_queries(w, 92, 2) + [t for t in not_queries(w, 93, 2) if '"' not in t] + ["t0 t1", "t5"]
    progs = [O.parse_query(t) for t in texts]
    w.ora.set_similarity(SIMS[sim])
    try:
        d, s, c, counts = run_scored(w, progs, 50, similarity=SIMS[sim])
        for i, t in enumerate(texts):
            docs, scores = w.ora.exec(progs[i], O.FLAG_ACCUM_SCORE)
            assert int(counts[i]) == len(docs), t
            td, ts = w.ora.topk(docs, scores, 50)
            # Trivial scores are small integers: ties everywhere, so the tie rule (docID ascending) is what is being checked
            assert d[i, : len(td)].tolist() == td.tolist(), (sim, t)
            np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0)
    finally:
        w.ora.set_similarity(0)


# ------------------------------------------------------------------------------------------ the default ("rich match") mode, SURVEY §8f-1
def rich_flat(docs, terms, present, freq, pos):
    """The canonical stream of oracle.to_exec_query_rich / ref_driver from the engine's arrays: per match  doc, nterms, then per
    matched term in ascending rank  rank, freq, pos[freq]."""
    out = []
    at = 0
    order = np.argsort(terms, kind="stable")
    for i, d in enumerate(docs.tolist()):
        offs = {}
        for k in range(len(terms)):  # positions are match-major, term-minor in the engine's term order
            offs[k] = at
            at += int(freq[i, k])
        ks = [k for k in order.tolist() if (int(present[i]) >> k) & 1]
        out += [d, len(ks)]
        for k in ks:
            f = int(freq[i, k])
            out += [int(terms[k]), f] + pos[offs[k] : offs[k] + f].tolist()
    assert at == len(pos)
    return np.array(out, dtype=np.uint32)


def run_rich(w, programs):
    b = w.T.Batch(w.ix, programs, w.T.FLAG_MATCHED_TERMS)
    b.run()
    b.sync()
    counts = b.counts()
    res = []
    for i in range(len(programs)):
        n = int(counts[i])
        docs = b.docset(i, n)
        res.append((docs,) + b.matched_terms(i, n))
    b.close()
    return res


@pytest.mark.parametrize("world,n", [("small", 12), ("dense", 12), ("medium", 5), ("small_l", 8)])
def test_rich_mode_matches_oracle(request, world, n):
    """exec_query's default mode: the matched terms of every match with their frequencies and positions, both codecs, every
    lowered query shape (conjunctions, unions, CNF, NOT, phrases)."""
    w = request.getfixturevalue(world)
    texts = template_queries(w, 101, n) + not_queries(w, 102, 2) + phrase_queries(w, 103, 2) + ["t0 t1", "t5", "t0 OR t1 OR t2"]
    progs = [O.parse_query(t) for t in texts]
    for t, p, (docs, terms, present, freq, pos) in zip(texts, progs, run_rich(w, progs)):
        wdocs, wflat, tt, ht = w.ora.exec_rich(p)
        assert np.array_equal(docs, wdocs), t
        got = rich_flat(docs, terms, present, freq, pos)
        assert int(freq.sum()) == ht and int(sum(bin(int(x)).count("1") for x in present)) == tt, t
        assert np.array_equal(got, wflat), t


def test_rich_mode_against_reference_fixtures(T, dev):
    """flags 0 records produced by the genuine reference (consider(const matched_document &) canonicalised by ref_driver)."""
    checked = 0
    for name in ("small", "dense"):
        g = json.load(open(os.path.join(GOLDEN, f"ref_{name}.json")))
        c = g["corpus"]
        w = World(T, dev, c["D"], c["V"], c["slots"], c["seed"])
        recs = [r for r in g["results"] if r["cmd"] == "query" and r["flags"] == 0 and gpu_lowers(r["q"], rich=True)]
        for r, (docs, terms, present, freq, pos) in zip(recs, run_rich(w, [O.parse_query(r["q"]) for r in recs])):
            assert len(docs) == r["n"] and str(O.fnv1a_docs(docs)) == r["fnv"], r["q"]
            assert int(freq.sum()) == r["hits_total"], r["q"]
            assert str(O.fnv1a_u32_stream(rich_flat(docs, terms, present, freq, pos))) == r["rich_fnv"], r["q"]
            checked += 1
        w.ix.close()
    assert checked >= 150


# ------------------------------------------------------------------------------------------ Optional (consttrueexpr under an AND)
OPT_TEMPLATES = ["t{a} <t{b}>", "t{a} t{b} <t{c} OR t{d}>", "t{a} <t{c}> t{b}", "(t{a} OR t{b}) <t{c}>", "t{a} <t{a}>", "(t{a} t{b} NOT t{e}) <t{c}>", '"t{a} t{b}" <t{c}>']


def opt_queries(w, seed, n):
    rows = w.T.gen_queries(w.V, seed, n, 5).tolist()
    head = [[0, 1, 2, 3, 4], [3, 0, 1, 4, 7], [5, 2, 0, 3, 9]]
    return [tpl.format(a=a, b=b, c=c, d=d, e=e) for a, b, c, d, e in head + rows for tpl in OPT_TEMPLATES]


@pytest.mark.parametrize("world", ["small", "dense", "small_l"])
def test_optional_matches_oracle(request, world):
    """`a <b>`: the documents of a; b adds its score where it matches (docset_iterators_scorers.cpp:77-104) and is reported among
    the matched terms where it matches (queryexec_ctx.cpp:418-432) — in all three execution modes."""
    w = request.getfixturevalue(world)
    texts = opt_queries(w, 111, 8)
    progs = [O.parse_query(t) for t in texts]
    sets, hashes, _ = run_docs_only(w, progs)
    for t, p, got in zip(texts, progs, sets):
        want, _ = w.ora.exec(p, O.FLAG_DOCUMENTS_ONLY)
        assert np.array_equal(got, want), t
    d, s, c, counts = run_scored(w, progs, 20)
    for i, t in enumerate(texts):
        docs, scores = w.ora.exec(progs[i], O.FLAG_ACCUM_SCORE)
        assert int(counts[i]) == len(docs), t
        td, ts = w.ora.topk(docs, scores, 20)
        assert d[i, : len(td)].tolist() == td.tolist(), t
        np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0)
    for t, p, (docs, terms, present, freq, pos) in zip(texts, progs, run_rich(w, progs)):
        wdocs, wflat, tt, ht = w.ora.exec_rich(p)
        assert np.array_equal(docs, wdocs) and np.array_equal(rich_flat(docs, terms, present, freq, pos), wflat), t


# ------------------------------------------------------------------------------------------ the reference-produced EDGE segment
def test_edge_segment_from_reference(T, dev):
    """tests/golden/ref_edge.json holds a segment the GENUINE reference wrote (raw index bytes + term table) with the codec's corner
    cases in it — hits with payloads of changing and constant length, a counted position-0 hit, documents of frequency 0, a
    document of 70000 hits (freq wraps in tokenpos_t), positions up to MaxPosition - 1, repeated positions — and the reference's
    answers.  The GPU reads bytes it did not write: decode, docID sets, BM25 (id, score) streams, phrases over payload-bearing
    hits, and the default mode's matched terms + positions."""
    import base64

    g = json.load(open(os.path.join(GOLDEN, "ref_edge.json")))
    index = np.frombuffer(base64.b64decode(g["index_b64"]), dtype=np.uint8)
    terms = np.array(g["terms"], dtype=np.uint32)
    ix = T.Index(dev, index, terms, g["docsCnt"])
    ora = O.Index.wrap(index, terms, g["docsCnt"], g["postings"], g["sumTermHits"])
    try:
        # codec seam
        df = terms[:, 0].astype(np.int64)
        docs, freqs, offs = ix.decode_terms(np.arange(len(terms), dtype=np.uint32), df)
        for r in g["results"]:
            if r["cmd"] == "decode":
                t = r["term"]
                d, f = docs[int(offs[t]) : int(offs[t + 1])], freqs[int(offs[t]) : int(offs[t + 1])]
                assert len(d) == r["n"] and str(O.fnv1a_docs(d)) == r["docs_fnv"], t
                assert str(O.fnv1a_docs(f & 0xFFFF)) == r["freqs_fnv"], t  # PostingsListIterator::freq is tokenpos_t
        assert int(freqs[int(offs[2]) : int(offs[3])].max()) == 70000
        # span seam: every fixture query, DocumentsOnly and the full (id, score) stream
        recs = [r for r in g["results"] if r["cmd"] == "queryfull"]
        for flags in (1, 2):
            rs = [r for r in recs if r["flags"] == flags]
            progs = [O.parse_query(r["q"]) for r in rs]
            b = T.Batch(ix, progs, T.FLAG_DOCUMENTS_ONLY if flags == 1 else T.FLAG_ACCUMULATED_SCORE, topk=0)
            b.run()
            b.sync()
            counts = b.counts()
            for i, r in enumerate(rs):
                got = b.docset(i, int(counts[i]))
                assert got.tolist() == r["docs"], (flags, r["q"])
                if flags == 2:
                    np.testing.assert_allclose(b.scores(i, int(counts[i])), r.get("scores", []), rtol=1e-5, atol=0)
            b.close()
            if flags == 2:  # and as top-K lists, planner's choice and the one-pass windows forced (frequency 0 scores 0, the wrapped one as wrapped)
                for opts in ({}, {"dense_min_postings": 0}):
                    with options(dev, **opts):
                        b = T.Batch(ix, progs, T.FLAG_ACCUMULATED_SCORE, topk=10)
                    b.run()
                    b.sync()
                    d, s, c = b.topk_results()
                    for i, r in enumerate(rs):
                        td, ts = ora.topk(np.array(r["docs"], dtype=np.uint32), np.array(r.get("scores", []), dtype=np.float64), 10)
                        assert int(b.counts()[i]) == r["n"] and d[i, : len(td)].tolist() == td.tolist(), (opts, r["q"])
                        np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0)
                    b.close()
        # default mode: matched terms, frequencies, positions (hits with payloads in between)
        w = type("W", (), {"T": T, "ix": ix})
        rich = [r for r in g["results"] if r["cmd"] == "query" and r["flags"] == 0]
        for r, (rdocs, rterms, present, freq, pos) in zip(rich, run_rich(w, [O.parse_query(r["q"]) for r in rich])):
            assert len(rdocs) == r["n"] and str(O.fnv1a_docs(rdocs)) == r["fnv"], r["q"]
            assert int(freq.sum()) == r["hits_total"], r["q"]
            assert str(O.fnv1a_u32_stream(rich_flat(rdocs, rterms, present, freq, pos))) == r["rich_fnv"], r["q"]
        assert len(rich) >= 8
    finally:
        ix.close()


def test_masked_documents_against_the_reference_s_filter_records(T, dev):
    """tests/golden/ref_masked.json: the reference with a rule-backed IndexDocumentsFilter — tested in the same condition as
    masked_documents_registry::test, right before consider() (exec.cpp:1095-1150 and the handlers of the other modes; matches.h:198-201).
    With the same documents installed through tri_index_set_masked the engine returns the reference's docID sets (DocumentsOnly), match
    counts + top-10 (AccumulatedScore: k_planes / k_fused / k_and + k_score all drop them before the ranking) and matched terms + hits
    (the default mode), for every record: 5 %, 30 % and 90 % of the documents dropped, two corpora."""
    g = json.load(open(os.path.join(GOLDEN, "ref_masked.json")))
    seen = {0: 0, 1: 0, 2: 0}
    for name, c in g["corpora"].items():
        w = World(T, dev, c["D"], c["V"], c["slots"], c["seed"])
        for fs, pm in g["filters"]:
            recs = [r for r in g["results"] if r["corpus"] == name and r["filter"] == [fs, pm]]
            with np.errstate(over="ignore"):
                w.ix.set_masked(O.masked_docs(c["D"], fs, pm))
            for flags in (1, 2, 0):
                rs = [r for r in recs if r["flags"] == flags]
                progs = [O.parse_query(r["q"], some_min=r["min"] or 1) for r in rs]
                if flags == 1:
                    sets, hashes, _ = run_docs_only(w, progs)
                    for r, got, h in zip(rs, sets, hashes):
                        assert len(got) == r["n"] and str(O.fnv1a_docs(got)) == r["fnv"] == str(int(h)), (name, fs, pm, r["q"])
                elif flags == 2:
                    for opts in ({}, {"dense_min_postings": 0}):  # the planner's choice; every eligible query in one pass
                        with options(dev, **opts):
                            d, s, cnt, counts = run_scored(w, progs, 10)
                        for i, r in enumerate(rs):
                            top = r.get("top", [])
                            assert int(counts[i]) == r["n"] and int(cnt[i]) == len(top), (name, fs, pm, r["q"])
                            assert d[i, : len(top)].tolist() == [x[0] for x in top], (name, fs, pm, r["q"], opts)
                            np.testing.assert_allclose(s[i, : len(top)], [x[1] for x in top], rtol=1e-5, atol=0)
                else:
                    for r, (docs, terms, present, freq, pos) in zip(rs, run_rich(w, progs)):
                        assert len(docs) == r["n"] and int(freq.sum()) == r["hits_total"], r["q"]
                        assert int(sum(bin(int(x)).count("1") for x in present)) == r["terms_total"], r["q"]
                        assert str(O.fnv1a_u32_stream(rich_flat(docs, terms, present, freq, pos))) == r["rich_fnv"], r["q"]
                seen[flags] += len(rs)
        w.ix.close()
    assert min(seen.values()) >= 400
