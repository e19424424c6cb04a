"""Full-size GPU parity (run with -m gpu on an MI355X): the 10M-document / 1M-term segment BASELINE.json's metric is quoted on.

Per QUERY, not in aggregate: match counts and FNV-1a of the docID set (DocumentsOnly; tri_batch_docset_hashes hashes the device
results without copying the sets back), top-K docID lists and scores (AccumulatedScore), against the CPU oracle on the same
segment bytes.  The samples cover both matching kernels of cfg2 (head x head pairs run as bitmap windows, Zipf pairs mostly as
candidate tiles), cfg4's phrases, and cfg3's 5-term mixes through the one-pass scored kernel on both codecs."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
D, V = 10_000_000, 1_000_000


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


@pytest.fixture(scope="module")
def google(T, dev):
    seg = T.Segment(D, V, 10, 42)
    ora = O.Index.wrap(seg.index, seg.terms, seg.docs_cnt, seg.sum_terms_docs, seg.sum_term_hits)
    ix = T.Index.from_segment(dev, seg)
    yield seg, ora, ix
    ix.close()


@pytest.fixture(scope="module")
def lucene(T, dev):
    seg = T.Segment(D, V, 10, 42, codec=2)
    ora = O.Index.generate(D, V, 10, 42, codec="lucene")
    ix = T.Index.from_segment(dev, seg)
    yield seg, ora, ix
    ix.close()


def check_docsets(T, ix, ora, progs, want_classes=()):
    b = T.Batch(ix, progs, T.FLAG_DOCUMENTS_ONLY)
    b.run()
    b.sync()
    counts, hashes, info = b.counts(), b.docset_hashes(), b.info()
    b.close()
    for name in want_classes:
        assert info[name] > 0, (name, info)
    total = 0
    for i, p in enumerate(progs):
        docs, _ = ora.exec(p, O.FLAG_DOCUMENTS_ONLY)
        assert int(counts[i]) == len(docs), (i, p.tolist(), int(counts[i]), len(docs))
        assert int(hashes[i]) == O.fnv1a_docs(docs), (i, p.tolist())
        total += len(docs)
    return total


def check_scored(T, ix, ora, progs, k, want_fused=True):
    b = T.Batch(ix, progs, T.FLAG_ACCUMULATED_SCORE, topk=k)
    b.run()
    b.sync()
    counts, (d, s, c), info = b.counts(), b.topk_results(), b.info()
    b.close()
    if want_fused:
        assert info["fused_queries"] + info["planes_queries"] > 0, info
    for i, p in enumerate(progs):
        docs, scores = ora.exec(p, O.FLAG_ACCUM_SCORE)
        assert int(counts[i]) == len(docs), (i, p.tolist(), int(counts[i]), len(docs))
        td, ts = ora.topk(docs, scores, k)
        assert int(c[i]) == len(td), (i, p.tolist())
        assert d[i, : len(td)].tolist() == td.tolist(), (i, p.tolist())
        np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0)


def test_cfg2_per_query_counts_and_hashes(T, google):
    """272 queries of the bench's own workload shape: 64 head x head pairs (TASK_PSET: plane algebra, millions of matches each),
    192 Zipf-drawn pairs of the cfg2 generator (mostly TASK_CAND: galloping / block-driven candidate tiles, plane probes) and 16
    TASK_DENSE ones (bitmap windows with decoded rows)."""
    from trinity_amd import workloads as W

    seg, ora, ix = google
    rng = np.random.default_rng(11)
    heads = [tuple(int(x) for x in rng.choice(24, 2, replace=False)) for _ in range(64)]
    progs = W.and2(heads) + W.and2(T.gen_queries(V, 1337, 16384, 2)[:192])
    # ... and 16 conjunctions of a head term with a union that holds a list too short for a plane: still bitmap windows with decoded rows (k_and_dense)
    for a, b, c in zip(rng.choice(24, 16), rng.integers(2000, 20000, 16), rng.choice(24, 16)):
        progs.append(np.array([T.tok(T.OP_TERM, int(a)), T.tok(T.OP_TERM, int(b)), T.tok(T.OP_OR, 2), T.tok(T.OP_TERM, int(c)), T.tok(T.OP_AND, 2)], dtype=np.uint32))
    total = check_docsets(T, ix, ora, progs, want_classes=("pset_queries", "dense_queries", "cand_queries"))
    assert total > 5_000_000


def test_cfg4_phrases_per_query(T, google):
    seg, ora, ix = google
    from trinity_amd import workloads as W

    progs, _, _, _, _ = W.build("cfg4", D, V, 10, 42, 96)
    assert check_docsets(T, ix, ora, progs) > 0


def test_cfg3_scored_lucene_per_query(T, lucene):
    """cfg3's four 5-term shapes on the lucene_codec segment, BM25 top-100: the one-pass scored kernel (unions, CNFs) and the
    candidate-tile + scoring path (sparse leads) at full size, plus the heaviest unions there are (the five head terms)."""
    from trinity_amd import workloads as W

    seg, ora, ix = lucene
    progs = W.mixed5(T.gen_queries(V, 1337, 8192, 5)[:64]) + [O.parse_query(t) for t in ("t0 OR t1 OR t2 OR t3 OR t4", "t0 t1 (t2 OR t3 OR t4)", "(t0 OR t1) (t2 OR t3) t4", "t0 t1 t2 t3 t4")]
    check_scored(T, ix, ora, progs, 100)


def test_cfg3_scored_google_per_query(T, google):
    seg, ora, ix = google
    from trinity_amd import workloads as W

    progs = W.mixed5(T.gen_queries(V, 1338, 8192, 5)[:32]) + [O.parse_query(t) for t in ("t0 OR t1 OR t2 OR t3 OR t4", "t0 t1")]
    check_scored(T, ix, ora, progs, 10)


def test_general_trees_full_size(T, google, lucene):
    """matchsome / NOT of a conjunction / AND under OR over head terms and sparse terms of the 10M-document segment: DocumentsOnly
    sets (many emitting tasks per query, each with its own output region) and BM25 top-100, per query against the oracle."""
    texts = [("[t0, t1, t2]", 2), ("[t0, t1, t2, t3, t4]", 3), ("t3 NOT (t0 t1)", 1), ("t2 OR (t0 t1)", 1), ("[t5, t900, t40000]", 2), ("t7 [t100, t2000, t30]", 2),
             ("(t0 NOT t1) OR (t2 NOT t3) OR t4", 1), ("t50000 OR (t1 t60000)", 1)]
    progs = [O.parse_query(q, some_min=mn) for q, mn in texts]
    seg, ora, ix = google
    check_docsets(T, ix, ora, progs)
    seg, ora, ix = lucene
    check_scored(T, ix, ora, progs, 100, want_fused=True)


def test_google_index_beyond_2_gib(T, dev):
    """codecs.h:26 allows chunks up to 4 GiB; until round 4 a google_codec index of 2 GiB or more was refused (bit 31 of a block's hits offset
    carried a flag).  A 34 M-document / 680 M-posting segment (2.4 GB): terms whose chunks lie beyond the 2 GiB mark — conjunctions, unions,
    phrases (k_phrase reads their hits), scores, and the default mode's positions (k_rich) — against the oracle on the same bytes."""
    seg = T.Segment(34_000_000, 1_000_000, 20, 42)
    assert seg.index.size >= (1 << 31) + (1 << 26), seg.index.size
    ora = O.Index.wrap(seg.index, seg.terms, seg.docs_cnt, seg.sum_terms_docs, seg.sum_term_hits)
    ix = T.Index.from_segment(dev, seg)
    far = np.nonzero(seg.terms[:, 1].astype(np.uint64) >= (1 << 31))[0]
    assert len(far) > 1000
    far = far[np.argsort(-seg.terms[far, 0].astype(np.int64), kind="stable")][:24].tolist()  # the longest lists out there
    texts = [f"t{a} t{b}" for a, b in zip(far[0::2], far[1::2])] + [f"t{a} OR t{b} OR t{c}" for a, b, c in zip(far[0::3], far[1::3], far[2::3])]
    texts += [f'"t{a} t{b}"' for a, b in zip(far[0::2], far[1::2])] + [f't1 OR "t{far[1]} t{far[2]}"', f"t{far[3]} NOT t{far[4]}"]
    texts += [f"t{k % 3} t{a}" for k, a in enumerate(far[:12])] + [f'"t{k % 3} t{a}"' for k, a in enumerate(far[:12])] + [f'"t{a} t{k % 3}"' for k, a in enumerate(far[:12])]  # (a head term's documents hold them)
    progs = [O.parse_query(t) for t in texts]
    b = T.Batch(ix, progs, T.FLAG_DOCUMENTS_ONLY)
    b.run()
    b.sync()
    counts, hashes = b.counts(), b.docset_hashes()
    nonempty = 0
    for i, (t, p) in enumerate(zip(texts, progs)):
        want, _ = ora.exec(p, O.FLAG_DOCUMENTS_ONLY)
        assert int(counts[i]) == len(want) and int(hashes[i]) == O.fnv1a_docs(want), t
        nonempty += len(want) > 0
    b.close()
    assert nonempty >= 24, nonempty
    b = T.Batch(ix, progs, T.FLAG_ACCUMULATED_SCORE, topk=10)
    b.run()
    b.sync()
    d, s, c = b.topk_results()
    for i, (t, p) in enumerate(zip(texts, progs)):
        docs, scores = ora.exec(p, O.FLAG_ACCUM_SCORE)
        td, ts = ora.topk(docs, scores, 10)
        assert d[i, : len(td)].tolist() == td.tolist(), t
        np.testing.assert_allclose(s[i, : len(td)], ts, rtol=1e-5, atol=0, err_msg=t)
    b.close()
    rich = progs[:4] + progs[-36::5]
    b = T.Batch(ix, rich, T.FLAG_MATCHED_TERMS)
    b.run()
    b.sync()
    counts = b.counts()
    for i, p in enumerate(rich):
        wdocs, wflat, tt, ht = ora.exec_rich(p)
        n = int(counts[i])
        docs = b.docset(i, n)
        terms, present, freq, pos = b.matched_terms(i, n)
        assert np.array_equal(docs, wdocs) and int(freq.sum()) == ht and int(sum(bin(int(x)).count("1") for x in present)) == tt
    b.close()
    ix.close()


def test_the_100k_mixed_batch_runs_whole_on_one_gpu():
    """`north_star`'s scaling batch — 100 K mixed queries (cfg5), strong scaling — at its N = 1 point: ONE GPU takes the whole batch (output regions of
    70 K DocumentsOnly queries, the plane rows, 30 K scored queries), through bench.py's own loop; per-query parity on a sample against the CPU oracle
    rides along, and the pipelined sets answer what a resident batch answers."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--scaling", "strong", "--workload", "cfg5", "--queries", "100000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "4",
                          "--rotating-sets", "0", "--delivered-steps", "0", "--scaling-ref-steps", "0"], capture_output=True, text=True, timeout=1200)  # fmt: skip
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["scaling"] == "strong" and out["n_gpus"] == 1 and out["config"]["queries_per_step"] == 100000 and out["config"]["workload"].startswith("cfg5")
    assert out["value"] > 0 and out["pipelined_results_equal_resident_batch"] and out["end_to_end"]["planning_included"]
    assert out["parity_check"]["equal"] and out["parity_check"]["queries"] >= 100 and out["parity_check"]["mismatches"] == 0
    # what the whole batch holds of the device while its loop runs (six sets of buffers alive — the resident diagnostic set, running, launched, read back, compiled,
    # being compiled — at 20 GB of bound-allocated output regions each, two indexes, their plane caches): stated, and within 60 % of the device (output
    # regions are bound-allocated by the lead's documents, not by a count pass — DESIGN.md §15.6)
    hbm = out["hbm_bytes_in_use"]
    assert 0 < hbm["engine_pool_in_use"] <= hbm["device_in_use"] < 0.6 * hbm["device_total"] and hbm["device_total"] > 200 << 30, hbm
