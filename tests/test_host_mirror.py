"""The host-side C++ operator surface (trinity_amd/csrc/host/trinity_gpu.hpp): compiles on CPU; on the GPU the
C++ driver tests/cpp/host_mirror_test.cpp runs Trinity-shaped application code (IndexSource, iterator trees,
exec_query with MatchedIndexDocumentsFilter / IndexDocumentsFilter / BM25 scorer, PostingsListIterator by hand)
and its output is compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O


@pytest.fixture(scope="module")
def T():
    import trinity_amd

    trinity_amd.build_all()
    return trinity_amd


def test_mirror_compiles_and_links(T):
    from trinity_amd.build import MIRROR_TEST_BIN

    assert os.path.exists(MIRROR_TEST_BIN)
    out = subprocess.run(["ldd", MIRROR_TEST_BIN], capture_output=True, text=True).stdout
    assert "libtrinity_hip.so" in out


@pytest.mark.gpu
@pytest.mark.parametrize("codec", ["GOOGLE", "LUCENE"])
def test_mirror_results_match_oracle(T, tmp_path, codec):
    from trinity_amd.build import MIRROR_TEST_BIN

    D, V = 20000, 500
    seg = T.Segment(D, V, 12, 7, codec=2 if codec == "LUCENE" else 1)
    ora = O.Index.generate(D, V, 12, 7)  # (the Google-coded oracle: the results do not depend on the codec the segment is stored in)
    ipath, tpath, hpath = str(tmp_path / "index"), str(tmp_path / "terms"), str(tmp_path / "hits.data")
    np.asarray(seg.index).tofile(ipath)
    np.ascontiguousarray(seg.terms, dtype=np.uint32).tofile(tpath)
    np.asarray(seg.hits).tofile(hpath)
    res = subprocess.run([MIRROR_TEST_BIN, ipath, tpath, str(D)] + (["LUCENE", hpath] if codec == "LUCENE" else []), capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.splitlines()[0] == "codec " + codec  # (Codecs::Lucene::AccessProxy / Codecs::Google::AccessProxy made the decoder)
    lines = {}
    for l in res.stdout.splitlines():
        k, _, rest = l.partition(" ")
        lines[k] = dict(kv.split("=") for kv in rest.split() if "=" in kv) or rest

    def expect(name, text, flags, keep=None, sim=0, some_min=1):
        ora.set_similarity(sim)
        docs, scores = ora.exec(O.parse_query(text, some_min=some_min), flags)
        ora.set_similarity(0)
        if keep is not None:
            m = keep(docs)
            docs = docs[m]
            scores = scores[m] if scores is not None else None
        r = lines[name]
        assert int(r["n"]) == len(docs), name
        assert int(r["fnv"]) == O.fnv1a_docs(docs), name
        if flags & 2:
            assert abs(float(r["score_sum"]) - scores.sum()) <= 1e-9 * max(1.0, scores.sum()), name

    expect("and_docs", "t0 t1", 1)
    expect("and_scored", "t0 t1", 2)
    expect("mixed_scored", "t0 t1 (t2 OR t3 OR t4)", 2)
    # DocsSetSpan::process(mp, min, max) as a windowed call (docset_spans.h:84, 292-296): the same matches and scores out of 8192-document windows,
    # in ascending order, with ONE tri_batch_create for the whole walk
    expect("span_windows", "t0 t1 (t2 OR t3 OR t4)", 2)
    assert lines["span_windows"]["batches"] == "1" and int(lines["span_windows"]["windows"]) >= 3 and lines["span_windows"]["ordered"] == "1" and lines["span_windows"]["end"] == "1"
    expect("or_even", "t3 OR t7", 1, keep=lambda d: (d & 1) == 0)
    expect("phrase_scored", '"t0 t1" t2', 2)
    expect("phrase_and_member_scored", '"t0 t1" t0', 2)  # ScorerWeights are per program token, not per term
    expect("two_phrases_scored", '"t0 t1" "t0 t2"', 2)
    expect("not_scored", "t3 t5 NOT (t1 OR t2)", 2)
    expect("optional_scored", "t3 t1 <t5 OR t2>", 2)
    expect("some_scored", "[t0, t1 t2, t3 OR t4]", 2, some_min=2)
    expect("some_docs", "[t0, t1 t2, t3 OR t4]", 1, some_min=2)
    expect("tree_scored", "t5 OR (t1 NOT (t2 t3))", 2)
    expect("tfidf_scored", "t0 t1 (t2 OR t3)", 2, sim=O.SIM_TFIDF)
    expect("trivial_scored", "t0 t1", 2, sim=O.SIM_TRIVIAL)
    expect("masked_scored", "t0 t1", 2, keep=lambda d: (d % 3) != 0)
    expect("masked_registry", "t0 t1", 2, keep=lambda d: (d % 3) != 0)  # exec_query(query, source, masked_documents_registry *, ...): exec.h:50
    expect("no_registry", "t0 t1", 1)
    expect("after_registry", "t0 t1", 1)  # the registry was per call
    expect("own_set_restored", "t0 t1", 1, keep=lambda d: (d % 5) != 0)  # ... and a call with a registry leaves the source's own set in place
    # default mode: the same digest from the oracle's canonical stream
    docs, flat, tt, ht = ora.exec_rich(O.parse_query("t0 t1 (t2 OR t3 OR t4)"))
    r = lines["rich"]
    assert (int(r["n"]), int(r["terms"]), int(r["hits"])) == (len(docs), tt, ht)
    M, digest, at, fl = (1 << 64) - 1, 1469598103934665603, 0, flat.tolist()
    while at < len(fl):
        doc, nt = fl[at], fl[at + 1]
        at += 2
        for _ in range(nt):
            rank, f = fl[at], fl[at + 1]
            x = 1469598103934665603
            for v in [doc, rank, f] + fl[at + 2 : at + 2 + f]:
                x = ((x ^ v) * 1099511628211) & M
            digest = (digest + x) & M
            at += 2 + f
    assert int(r["digest"]) == digest
    assert int(lines["unknown"]["n"]) == 0
    expect("batch0", "t1 t2", 1)
    expect("batch1", "t8 OR t9", 1)
    # codec seam driven by hand
    d5, f5 = ora.decode_term(5)
    h = 1469598103934665603
    for d, f in zip(d5.tolist(), f5.tolist()):
        h = ((h ^ ((d * 31 + f) & 0xFFFFFFFFFFFFFFFF)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    p = lines["pli"]
    assert int(p["n"]) == len(d5) and int(p["h"]) == h
    i = int(np.searchsorted(d5, 1000))
    assert int(p["adv1000"]) == d5[i] and int(p["freq"]) == f5[i] and int(p["again"]) == d5[i]
    assert "invalid_argument" in res.stdout.splitlines()[-1]


def test_workload_generators_on_cpu(T):
    """trinity_amd/workloads.py (what bench.py --workload times): every program parses in the oracle, and the phrases that
    csrc/host/synth.cpp samples from a document's consecutive token slots — by addressing the corpus stream as a counter-based
    generator, without materialising the corpus — really occur in the oracle's independently generated corpus."""
    from trinity_amd import workloads as W

    D, V = 5000, 500
    ora = O.Index.generate(D, V, 10, 42)
    for name in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5"):
        parts, desc = W.build_parts(name, D, V, 10, 42, 60)
        assert sum(len(pt.programs) for pt in parts) == 60 and desc.startswith(name)
        assert (len(parts) == 2 and parts[1].flags & 2 and parts[1].topk == 100) if name == "cfg5" else len(parts) == 1  # cfg5's 30 % share is scored
        hits = 0
        for pt in parts:
            for p in pt.programs:
                docs, _ = ora.exec(p, O.FLAG_DOCUMENTS_ONLY if not (pt.flags & 2) else O.FLAG_ACCUM_SCORE)
                hits += len(docs) > 0
        if name == "cfg4":
            # even rows of each half are document-sampled: at least those must match
            assert hits >= 30


def test_write_side_mirror_compiles_and_links(T, tmp_path):
    """trinity_gpu_write.hpp — SegmentIndexSession (begin / insert / commit) and the dictionary-wide merge over tri_commit_* / tri_merge_google — and the
    application-shaped driver tests/cpp/host_mirror_write_test.cpp build warning-free and link against the engine (the driver needs a device to run)."""
    from trinity_amd.build import PKG, ROOT

    exe = str(tmp_path / "host_mirror_write_test")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_mirror_write_test.cpp"), "-L" + PKG, "-ltrinity_hip",
                        "-Wl,-rpath," + PKG], capture_output=True, text=True)  # fmt: skip
    assert r.returncode == 0, r.stderr[-3000:]
    assert "libtrinity_hip.so" in subprocess.run(["ldd", exe], capture_output=True, text=True).stdout


@pytest.mark.gpu
def test_write_side_driver_commits_and_merges_on_the_device(T):
    """tests/cpp/host_mirror_write_test: application code against the mirror's SegmentIndexSession (begin / insert / commit) and merge_google, RUN on the
    device — three documents fed out of order, a payload, a hit at position 0 (counted, never stored).  The committed bytes and term table equal the
    host encoder's over commit's walk (the driver checks it and says so), the statistics count the dropped hit, and the committed segment uploaded
    and merged with itself comes back unchanged."""
    from trinity_amd.build import MIRROR_WRITE_BIN

    res = subprocess.run([MIRROR_WRITE_BIN], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    out = res.stdout.splitlines()
    head = out[0].replace(",", "").split()  # index N bytes T terms documents D postings P hits H
    assert head[0] == "index" and int(head[3]) == 9 and int(head[6]) == 3 and int(head[8]) == 11 and int(head[10]) == 13, out[0]  # (12 stored hits + the one at position 0)
    table = {l.split()[0]: l for l in out[1:10]}
    assert set(table) == {"world", "of", "warcraft", "mists", "hello", "again", "mice", "and", "men"}
    assert "documents=2" in table["world"] and "documents=2" in table["of"] and "documents=1" in table["mice"]
    assert any(l.startswith("host_encoder") and "same=1" in l and "term_table_same=1" in l for l in out), res.stdout
    assert out[-1].startswith("merged") and out[-1].endswith("same=1"), out[-1]
