#!/usr/bin/env python3
"""Generate tests/golden/*.json from the GENUINE reference (oracle/_ref/ref_driver, built from
/root/reference by oracle/Makefile).  Runs only in the build container (the GPU box has no reference).

The fixtures are DATA: inputs (corpus parameters of this repo's deterministic generator, query texts,
seeds) and the reference's outputs (counts, FNV-1a hashes, first/last docIDs, score sums, top-K, and full
(docID, score) lists on the small corpus).  No reference source text is stored.

usage: python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

CORPORA = {
    # name: (D, V, slots, seed)
    "tiny": (2000, 200, 10, 42),
    "small": (20000, 2000, 10, 42),
    "dense": (20000, 500, 12, 7),  # heavy duplication inside documents (freq > 1), long lists
}

TEMPLATES = [
    "t{a} t{b}",
    "t{a} t{b} t{c} t{d} t{e}",
    "t{a} OR t{b}",
    "t{a} OR t{b} OR t{c} OR t{d} OR t{e}",
    "t{a} t{b} (t{c} OR t{d} OR t{e})",
    "(t{a} OR t{b}) (t{c} OR t{d}) t{e}",
    '"t{a} t{b}"',
    '"t{a} t{b} t{c}"',
    '"t{a} t{b}" t{c}',
    # logicalnot -> DocsSetIterators::Filter / FilteredDocsSetSpan (exec.cpp:424-427, 488-501)
    "t{a} NOT t{b}",
    "t{a} t{b} NOT t{c}",
    # (a root-level `(x OR y) NOT z` is left out on purpose: when cost(z) <= cost(x OR y) the reference runs it as
    #  FilteredDocsSetSpan over DocsSetSpanForDisjunctions, whose process() ignores `min` (docset_spans.cpp:98-110), so the
    #  excluded documents are emitted anyway — 1902 instead of 1054 documents for `(t0 OR t1) NOT t2` on the tiny corpus.
    #  The same query under an AND takes the iterator path and is exact.)
    "(t{a} OR t{b}) t{d} NOT t{c}",
    "t{a} NOT (t{b} OR t{c})",
    "t{a} NOT (t{b} t{c})",
    # consttrueexpr under an AND -> DocsSetIterators::Optional (exec.cpp:366-377; parser flag ParseConstTrueExpr)
    "t{a} <t{b}>",
    "t{a} t{b} <t{c} OR t{d}>",
    "t{a} <t{c}> t{b}",
    "(t{a} OR t{b}) <t{c}>",
]


def commands_for(name, D, V, slots, seed):
    cmds = ["index"]
    terms = [0, 1, 2, 3, 7, 19, V // 4, V // 2, V - 1]
    for t in terms:
        cmds.append(f"decode {t}")
    for t in (0, 1, 5, 19):
        for s in (1, 2, 3):
            cmds.append(f"advance {t} {s} 4000")
    for t in (0, 3, V // 4):
        cmds.append(f"positions {t} 3")
    qs = O.gen_queries(V, 1337, 12, 5)
    head = [[0, 1, 2, 3, 4], [1, 0, 2, 5, 9], [3, 7, 11, 0, 2]]
    for ri, row in enumerate(head + qs.tolist()):
        full = name == "tiny" and ri < 3  # full (docID, score) lists only for the three head rows
        a, b, c, d, e = [int(x) for x in row]
        for tpl in TEMPLATES:
            text = tpl.format(a=a, b=b, c=c, d=d, e=e)
            for flags in (1, 2):
                if full:
                    cmds.append(f"queryfull {flags} {text}")
                else:
                    cmds.append(f"query {flags} {10 if flags == 2 else 0} {text}")
    # the default ("rich match") mode, flags 0: matched terms + hits per document, canonicalised by the driver (rich_fnv)
    for ri, row in enumerate(head + qs.tolist()[:6]):
        a, b, c, d, e = [int(x) for x in row]
        for tpl in TEMPLATES:
            cmds.append("query 0 0 " + tpl.format(a=a, b=b, c=c, d=d, e=e))
    # matchsome -> DisjunctionSome (exec.cpp:276-283): `[a, b, ...]` with the threshold set on the node; all three modes
    SOME = [("[t{a}, t{b}, t{c}]", 2), ("[t{a}, t{b}, t{c}, t{d}, t{e}]", 2), ("[t{a}, t{b}, t{c}, t{d}, t{e}]", 3), ("[t{a}, t{b}, t{c}, t{d}, t{e}]", 5),
            ("t{a} [t{b}, t{c}, t{d}]", 2), ("[t{a}, t{b} t{c}, t{d} OR t{e}]", 2)]
    for ri, row in enumerate(head + qs.tolist()[:7]):
        a, b, c, d, e = [int(x) for x in row]
        for tpl, mn in SOME:
            text = tpl.format(a=a, b=b, c=c, d=d, e=e)
            cmds.append(f"querysome 1 0 {mn} {text}")
            cmds.append(f"querysome 2 10 {mn} {text}")
            cmds.append(f"querysome 0 0 {mn} {text}")
    # the other two scorers of similarity.h (TF-IDF :75-163, Trivial :56-72) on a subset: scored records carry "sim"
    SIM_TEMPLATES = ["t{a} t{b}", "t{a} OR t{b} OR t{c}", "t{a} t{b} (t{c} OR t{d} OR t{e})", '"t{a} t{b}" t{c}', "t{a} t{b} NOT t{c}"]
    for sim in ("tfidf", "trivial"):
        cmds.append(f"sim {sim}")
        for ri, row in enumerate(head + qs.tolist()[:5]):
            a, b, c, d, e = [int(x) for x in row]
            for tpl in SIM_TEMPLATES:
                text = tpl.format(a=a, b=b, c=c, d=d, e=e)
                cmds.append(f"queryfull 2 {text}" if name == "tiny" and ri < 2 else f"query 2 10 {text}")
    cmds.append("sim bm25")
    # several phrases that share terms (the candidate document's DocWordsSpace and term hits are shared by all phrases of the query,
    # queryexec_ctx.cpp:317-351).  AccumulatedScore and the default mode only: in DocumentsOnly mode the reference itself returns a
    # strict SUBSET of the documents its other two modes return for these queries (e.g. 4 instead of 22 for `"t0 t1" "t1 t2"` on the
    # tiny corpus) — a reference defect the fixtures steer around
    OVERLAP = ['"t{a} t{b}" "t{b} t{c}"', '"t{a} t{b}" "t{c} t{a}"', '"t{a} t{b}" "t{a} t{c}"', '"t{a} t{b} t{c}" "t{b} t{c}"', '"t{a} t{b}" t{a}']
    for a, b, c in [(0, 1, 2), (1, 0, 2), (2, 1, 0), (0, 2, 1), (3, 1, 0), (1, 2, 3)]:
        for tpl in OVERLAP:
            text = tpl.format(a=a, b=b, c=c)
            cmds.append(f"query 2 10 {text}")
            cmds.append(f"query 0 0 {text}")
    # a term with zero documents (if the corpus has one) and an out-of-vocabulary term
    cmds.append(f"query 1 0 t0 t{V + 5}")
    cmds.append(f"query 1 0 t0 OR t{V + 5}")
    return cmds


EDGE_QUERIES = ["t0 t1", "t1", "t2", "t2 t0", "t0 OR t1 OR t6", '"t4 t5"', '"t5 t4"', '"t4 t5" t0', "t3 t4", "t7", "t0 NOT t1", "t1 <t6>", "t3 OR t7", "t2 t3",
                "t1 t3", "t6", "t4 t5", "t0 t3 t7", '"t4 t5" t3']
# (term 2 holds the document with 70000 hits: the reference's own hit walks lose their place behind it — materialize_hits reads
#  freq as tokenpos_t and trips its payload-size assertion —, so the fixtures keep it out of phrases and of the default mode and
#  pin what IS defined: its docIDs, and the wrapped frequency that scores.)
EDGE_RICH = ["t0 t1", '"t4 t5" t0', "t3 t7", "t1 t3", "t0 t3 t7", "t3 OR t7", "t1 <t6>", '"t4 t5" t3']


def edge_fixture():
    """tests/golden/ref_edge.json: the edge corpus as REFERENCE-PRODUCED bytes (raw `index` + term table, base64) and the reference's
    answers on it — so the oracle and the GPU read bytes they did not write: payloads, position 0, freq 0, wrapped freq, MaxPosition."""
    import base64
    import tempfile

    import numpy as np

    with tempfile.TemporaryDirectory() as td:
        prefix = os.path.join(td, "edge")
        cmds = ["index", f"dumpindex {prefix}"] + [f"decode {t}" for t in range(8)] + [f"hits {t}" for t in (0, 1, 3, 4, 5, 6, 7)]
        for t in (0, 1, 3, 4):
            cmds += [f"advance {t} {s} 2000" for s in (1, 2)]
        for q in EDGE_QUERIES:
            cmds += [f"queryfull 1 {q}", f"queryfull 2 {q}"]
        cmds += [f"query 0 0 {q}" for q in EDGE_RICH]
        res = O.run_ref_driver_edge(cmds)
        assert len(res) == len(cmds), (len(res), len(cmds))
        index = open(prefix + ".index", "rb").read()
        terms = np.fromfile(prefix + ".terms", dtype=np.uint32).reshape(-1, 3)
    out = {"corpus": "edge (oracle/ref_driver.cpp edge_hits)", "docsCnt": res[0]["docsCnt"], "sumTermHits": res[0]["sumTermHits"], "postings": res[0]["postings"],
           "index_b64": base64.b64encode(index).decode(), "terms": terms.tolist(), "results": res}  # fmt: skip
    path = os.path.join(HERE, "ref_edge.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(path, len(res), "records", os.path.getsize(path), "bytes")


TREE_EXTRA = ["t{a} t{a} t{b}", "t{a} OR (t{b} t{c})", "t{a} (t{b} OR (t{c} t{d}))", "(t{a} NOT t{b}) OR (t{c} NOT t{d}) OR t{e}", "t{a} <t{b} t{c}>", "t{a} NOT (t{b} OR (t{c} t{d}))",
              "(t{a} t{b}) OR (t{c} t{d})", "t{a} OR t{b} OR (t{c} t{d} t{e})", "t{a} t{b} t{c} NOT (t{d} t{e})"]
SOME_TEMPLATES = [("[t{a}, t{b}, t{c}]", 2), ("[t{a}, t{b}, t{c}, t{d}, t{e}]", 2), ("[t{a}, t{b}, t{c}, t{d}, t{e}]", 3), ("[t{a}, t{b}, t{c}, t{d}, t{e}]", 5),
                  ("t{a} [t{b}, t{c}, t{d}]", 2), ("[t{a}, t{b} t{c}, t{d} OR t{e}]", 2)]


def tree_fixture():
    """tests/golden/ref_trees.json: what the reference's own compile_query makes of the fixture queries — the exec_node trees
    exec.cpp:253-449 builds its iterators from (ref_driver `tree`), next to the reference's answers for the same queries on the tiny
    corpus.  tests lower the TREES to the C-ABI's postfix programs (oracle_lib.program_from_exec_tree), so the planner is driven by
    real compiler output — reordered, de-duplicated, runs of terms collapsed — rather than by this repo's own query parser."""
    name = "tiny"
    D, V, slots, seed = CORPORA[name]
    rows = [[0, 1, 2, 3, 4], [1, 0, 2, 5, 9], [3, 7, 11, 0, 2], [19, 4, 0, 50, 1], [7, 5, 3, 1, 0]]
    cmds = []
    for a, b, c, d, e in rows:
        for tpl in TEMPLATES + TREE_EXTRA:
            if '"' in tpl and tpl.count('"') > 2:
                continue
            text = tpl.format(a=a, b=b, c=c, d=d, e=e)
            cmds += [f"tree 0 {text}", f"query 1 0 {text}", f"query 2 10 {text}"]
        for tpl, mn in SOME_TEMPLATES:
            text = tpl.format(a=a, b=b, c=c, d=d, e=e)
            cmds += [f"tree {mn} {text}", f"querysome 1 0 {mn} {text}", f"querysome 2 10 {mn} {text}"]
    res = O.run_ref_driver(D, V, slots, seed, cmds)
    assert len(res) == len(cmds), (len(res), len(cmds))
    recs = []
    for i in range(0, len(res), 3):
        t, r1, r2 = res[i : i + 3]
        assert t["cmd"] == "tree" and t["q"] == r1["q"] == r2["q"]
        recs.append({"q": t["q"], "min": t["min"], "tree": t["tree"], "n": r1["n"], "fnv": r1["fnv"], "score_sum": r2["score_sum"], "top": r2.get("top", [])})
    out = {"corpus": {"D": D, "V": V, "slots": slots, "seed": seed}, "results": recs}
    path = os.path.join(HERE, "ref_trees.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(path, len(recs), "records", os.path.getsize(path), "bytes")


def random_query(rng, pool):
    """A random, fully parenthesised query text over distinct terms of `pool` (<= 8 of them): AND / OR / NOT / <optional> / [matchsome]."""

    it = iter(rng.sample(pool, 8))

    def split(budget, n):
        parts = [1] * n
        for _ in range(budget - n):
            parts[rng.randrange(n)] += 1
        return parts

    def node(depth, budget):
        if depth == 0 or budget == 1 or rng.random() < 0.2:
            return f"t{next(it)}"
        kind = rng.choice(["and", "and", "or", "or", "not", "some", "opt"])
        if kind == "some" and budget >= 3:
            n = min(rng.randint(3, 4), budget)
            return "[" + ", ".join(node(depth - 1, b) for b in split(budget, n)) + "]"
        if kind in ("not", "opt"):
            a, b = split(budget, 2)
            return "(" + node(depth - 1, a) + (" NOT " + node(depth - 1, b) if kind == "not" else " <" + node(depth - 1, b) + ">") + ")"
        n = min(rng.randint(2, 3), budget)
        return "(" + (" OR " if kind == "or" else " ").join(node(depth - 1, b) for b in split(budget, n)) + ")"

    q = node(3, rng.randint(3, 8))
    return q[1:-1] if q.startswith("(") and q.endswith(")") else q


def random_fixture():
    """tests/golden/ref_random.json: 400 random query trees over the tiny corpus — the reference's compiled exec_node tree of each and its
    answers in all three modes.  Left out: the queries the reference aborts or hangs on (a matchsome nested in a matchsome with a
    threshold above 1), and root-level `(x OR ...) NOT z` / `[...] NOT z`, where its DocumentsOnly / AccumulatedScore spans emit the
    excluded documents while its own default mode does not (FilteredDocsSetSpan over a disjunction span: DESIGN.md §8)."""
    import random
    import subprocess

    rng = random.Random(20260926)
    D, V, slots, seed = CORPORA["tiny"]
    qs = []
    while len(qs) < 400:
        q = random_query(rng, list(range(40)))
        if " " in q:
            qs.append((q, rng.choice([1, 2, 2, 3]) if "[" in q else 0))

    def cmds_of(q, mn):
        return [f"tree {mn} {q}"] + ([f"querysome {f} {k} {mn} {q}" for f, k in ((1, 0), (2, 10), (0, 0))] if mn else [f"query {f} {k} {q}" for f, k in ((1, 0), (2, 10), (0, 0))])

    def run(chunk):
        cmds = sum((cmds_of(q, mn) for q, mn in chunk), [])
        out = subprocess.run([O.REF_DRIVER, str(D), str(V), str(slots), str(seed)], input="\n".join(cmds) + "\n", capture_output=True, text=True, check=True, timeout=120)
        res = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(res) == len(cmds)
        return [res[i : i + 4] for i in range(0, len(res), 4)]

    recs, dropped = [], []
    for i in range(0, len(qs), 16):
        chunk = qs[i : i + 16]
        try:
            got = run(chunk)
        except (subprocess.CalledProcessError, subprocess.TimeoutExpired, AssertionError):
            got = []
            for one in chunk:  # a query the reference dies on takes its chunk with it: one by one
                try:
                    got += run([one])
                except (subprocess.CalledProcessError, subprocess.TimeoutExpired, AssertionError):
                    got.append(None)
                    dropped.append(one[0])
        for (q, mn), r in zip(chunk, got):
            if r is None:
                continue
            t, r1, r2, r0 = r
            root = t["tree"]
            if root["op"] == "not" and root["k"][0]["op"] in ("or", "anyterms", "some"):
                dropped.append(q)
                continue
            recs.append({"q": q, "min": mn, "tree": root, "n": r1["n"], "fnv": r1["fnv"], "score_sum": r2["score_sum"], "top": r2.get("top", []),
                         "rich_fnv": r0["rich_fnv"], "terms_total": r0["terms_total"], "hits_total": r0["hits_total"]})  # fmt: skip
    out = {"corpus": {"D": D, "V": V, "slots": slots, "seed": seed}, "dropped": dropped, "results": recs}
    path = os.path.join(HERE, "ref_random.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(path, len(recs), "records", len(dropped), "dropped", os.path.getsize(path), "bytes")


# A multi-word phrase under an OR, inside a matchsome, on the excluded side of a NOT, as the optional side: DocsSetIterators::Phrase is an
# iterator like any other (exec.cpp:253-449, docset_iterators.cpp:66-224).  AccumulatedScore and the default mode only: in DocumentsOnly
# mode the reference crashes on a Phrase inside a root OR.
PHRASE_TREES = [('"t{a} t{b}" OR t{c}', 0), ('t{c} OR "t{a} t{b}" OR "t{d} t{e}"', 0), ('t{e} ("t{a} t{b}" OR t{c})', 0), ('t{d} NOT ("t{a} t{b}" t{c})', 0),
                ('t{d} NOT "t{a} t{b}"', 0), ('t{e} OR ("t{a} t{b}" t{c})', 0), ('t{d} <"t{a} t{b}">', 0), ('("t{a} t{b}" OR "t{b} t{c}") t{d}', 0),
                ('"t{a} t{b} t{c}" OR (t{d} t{e})', 0), ('["t{a} t{b}", t{c}, t{d}]', 2), ('["t{a} t{b}", "t{c} t{d}", t{e}]', 1), ('t{e} ["t{a} t{b}", t{c}, t{d}]', 2)]


def phrase_tree_fixture():
    """tests/golden/ref_phrase_trees.json: the reference's compiled tree and its answers (top-10 + score sum, and the default mode's matched
    terms and hits) for phrases inside trees, on the tiny and the dense corpus.  The GPU planner does not lower these shapes yet (it leaves
    such a query out with TRI_ERR_UNSUPPORTED): the fixture pins the oracle for the day it does."""
    import subprocess

    out = {"corpora": {}, "results": []}
    for name in ("tiny", "dense"):
        D, V, slots, seed = CORPORA[name]
        out["corpora"][name] = {"D": D, "V": V, "slots": slots, "seed": seed}
        rows = [[0, 1, 2, 3, 4], [1, 0, 2, 5, 9], [3, 7, 11, 0, 2], [2, 3, 0, 1, 6]] + O.gen_queries(V, 1337, 6, 5).tolist()
        for row in rows:
            a, b, c, d, e = [int(x) for x in row]
            for tpl, mn in PHRASE_TREES:
                q = tpl.format(a=a, b=b, c=c, d=d, e=e)
                cmds = [f"tree {mn} {q}"] + ([f"querysome 2 10 {mn} {q}", f"querysome 0 0 {mn} {q}"] if mn else [f"query 2 10 {q}", f"query 0 0 {q}"])
                try:
                    got = subprocess.run([O.REF_DRIVER, str(D), str(V), str(slots), str(seed)], input="\n".join(cmds) + "\n", capture_output=True, text=True, check=True, timeout=60)
                    res = [json.loads(l) for l in got.stdout.splitlines() if l.startswith("{")]
                    assert len(res) == 3
                except (subprocess.CalledProcessError, subprocess.TimeoutExpired, AssertionError):
                    out.setdefault("dropped", []).append([name, q])
                    continue
                t, r2, r0 = res
                rec = {"corpus": name, "q": q, "min": mn, "tree": t["tree"], "n": r2["n"], "score_sum": r2["score_sum"], "top": r2.get("top", []),
                       "rich_n": r0["n"], "terms_total": r0["terms_total"], "hits_total": r0["hits_total"], "rich_fnv": r0["rich_fnv"]}  # fmt: skip
                # By rule, not by result: where a phrase can FAIL on a document that the query still matches while its terms hold the document —
                # the optional side, a matchsome with a threshold above 1 — the reference's default mode reports those terms with the right
                # frequency but the hit array of whatever document they were last materialised for (positions that do not ascend, position 0 of
                # a payload-less hit: `t3 <"t0 t1">` on the tiny corpus, 20 of 689 documents).  The hash of that stream is left out; the
                # match count, the term and hit totals and the scored answers stay.
                if '<"' in q or (mn >= 2 and '"' in q):
                    rec["rich_fnv"] = None
                out["results"].append(rec)
    path = os.path.join(HERE, "ref_phrase_trees.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(path, len(out["results"]), "records", len(out.get("dropped", [])), "dropped", os.path.getsize(path), "bytes")


MASKED_TEMPLATES = ["t{a} t{b}", "t{a} t{b} t{c} t{d} t{e}", "t{a} OR t{b}", "t{a} OR t{b} OR t{c} OR t{d} OR t{e}", "t{a} t{b} (t{c} OR t{d} OR t{e})",
                    "(t{a} OR t{b}) (t{c} OR t{d}) t{e}", '"t{a} t{b}"', '"t{a} t{b}" t{c}', "t{a} NOT t{b}", "t{a} <t{b}>", "t{a} NOT (t{b} t{c})", "t{a} OR (t{b} t{c})"]


def masked_fixture():
    """tests/golden/ref_masked.json: documents ruled out right before consider().  The reference drops a match when
    masked_documents_registry::test(id) or IndexDocumentsFilter::filter(id) says so, in the same `if` of the same handler, in every execution
    mode (exec.cpp:1095-1150 and the handlers after it; matches.h:198-201).  The registry's own structures cannot be built here
    (docidupdates.cpp needs boost), the filter hook can: `filter <seed> <permille>` hands exec_query a rule-backed IndexDocumentsFilter
    (document d is dropped when splitmix64(seed + d) % 1000 < permille — masked_docs() below rebuilds the set) and the records pin what
    comes out: docID sets (flags 1), scores / top-10 (flags 2: the dropped documents leave the top-K and the score sum), matched terms
    and hits (flags 0).  The oracle and the GPU reproduce them with that set installed as the segment's masked documents."""
    out = {"corpora": {}, "filters": [[7, 300], [11, 50], [3, 900]], "results": []}
    for name in ("tiny", "dense"):
        D, V, slots, seed = CORPORA[name]
        out["corpora"][name] = {"D": D, "V": V, "slots": slots, "seed": seed}
        rows = [[0, 1, 2, 3, 4], [1, 0, 2, 5, 9], [3, 7, 11, 0, 2]] + O.gen_queries(V, 1337, 4, 5).tolist()
        cmds = []
        for fs, pm in out["filters"]:
            cmds.append(f"filter {fs} {pm}")
            for row in rows:
                a, b, c, d, e = [int(x) for x in row]
                for tpl in MASKED_TEMPLATES:
                    q = tpl.format(a=a, b=b, c=c, d=d, e=e)
                    cmds += [f"query 1 0 {q}", f"query 2 10 {q}", f"query 0 0 {q}"]
            cmds.append("querysome 2 10 2 [t0, t1, t2, t3]")
            cmds.append("querysome 1 0 2 [t0, t1, t2, t3]")
        res = [r for r in O.run_ref_driver(D, V, slots, seed, cmds) if r["cmd"] != "filter"]
        for r in res:
            assert r["filter"][1] > 0
            r.pop("rich_docs", None)  # (the full per-document dumps are a debugging aid; the hash covers them)
            r["corpus"] = name
        out["results"] += res
    path = os.path.join(HERE, "ref_masked.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(path, len(out["results"]), "records", os.path.getsize(path), "bytes")


def merge_fixture():
    """tests/golden/ref_merge.json: Codecs::Google::IndexSession::merge (google_codec.cpp:186-438), driven per term as MergeCandidatesCollection::merge
    does (merge.cpp:254-288: begin_term / merge(the participants that hold documents, most recent first) / end_term) over small segments written by
    the reference's own encoder.  Each record carries the participants' INPUT postings (documents, positions, payloads) and the chunk bytes the
    reference wrote per output term: what pins "the union of the documents, a document from the most recent participant that holds it, its hits and
    payloads carried over, re-blocked by the encoder".  The masked registries are empty (see the `merge` command in oracle/ref_driver.cpp)."""
    cases = [(11, 3, 36, 3000), (5, 4, 24, 700), (3, 1, 12, 500), (17, 2, 16, 120)]  # (seed, participants, terms, documentIDs below): fewer IDs -> more shared documents
    res = O.run_ref_driver(1000, 100, 10, 42, [f"merge {a} {b} {c} {d}" for a, b, c, d in cases])
    assert len(res) == len(cases) and all(r["out"] for r in res)
    path = os.path.join(HERE, "ref_merge.json")
    with open(path, "w") as f:
        json.dump({"cases": [list(c) for c in cases], "results": res}, f, separators=(",", ":"))
    print(path, len(res), "records", os.path.getsize(path), "bytes")


def commit_fixture():
    """tests/golden/ref_commit.json: SegmentIndexSession::commit (indexer.cpp:311-560) of the genuine reference over sessions fed out of (term, document)
    order — the input in insertion order with the session's transient term ids, and the `index` bytes + per-term chunks commit wrote (`ref_driver commit`;
    the commit runs in a child process that dies at the one call this image cannot link, after the files are written: see the command's comment).
    Pins the walk tri_commit_google reproduces: buckets by termID & 31 in bucket order, (termID, documentID) inside a bucket, hits replayed."""
    cases = [(3, 400, 60, 20000), (9, 1500, 300, 70000)]  # (seed, documents, vocabulary, documentIDs below)
    res = O.run_ref_driver(1000, 100, 10, 42, [f"commit {a} {b} {c} {d}" for a, b, c, d in cases])
    assert len(res) == len(cases) and all(r["terms"] and r["index"] for r in res)
    path = os.path.join(HERE, "ref_commit.json")
    with open(path, "w") as f:
        json.dump({"cases": [list(c) for c in cases], "results": res}, f, separators=(",", ":"))
    print(path, len(res), "records", os.path.getsize(path), "bytes")


def main():
    if "--commit-only" in sys.argv:
        return commit_fixture()
    if "--merge-only" in sys.argv:
        return merge_fixture()
    if "--phrase-trees-only" in sys.argv:
        return phrase_tree_fixture()
    if "--trees-only" in sys.argv:
        return tree_fixture()
    if "--random-only" in sys.argv:
        return random_fixture()
    if "--masked-only" in sys.argv:
        return masked_fixture()
    edge_fixture()
    tree_fixture()
    random_fixture()
    phrase_tree_fixture()
    masked_fixture()
    merge_fixture()
    commit_fixture()
    for name, (D, V, slots, seed) in CORPORA.items():
        cmds = commands_for(name, D, V, slots, seed)
        res = O.run_ref_driver(D, V, slots, seed, cmds)
        assert len(res) == len(cmds), (len(res), len(cmds))
        out = {"corpus": {"D": D, "V": V, "slots": slots, "seed": seed}, "query_seed": 1337, "results": res}
        path = os.path.join(HERE, f"ref_{name}.json")
        with open(path, "w") as f:
            json.dump(out, f, separators=(",", ":"))
        print(path, len(res), "records", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
