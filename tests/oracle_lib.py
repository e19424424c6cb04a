"""ctypes binding of the CPU oracle (oracle/_ref/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (trinity_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.environ.get("TRINITY_ORACLE_LIB") or os.path.join(ORACLE_DIR, "_ref", "liboracle.so")  # (the env override: the sanitizer build, tests/test_oracle_sanitized.py)
REF_DRIVER = os.path.join(ORACLE_DIR, "_ref", "ref_driver")

DOCIDS_END = 0xFFFFFFFF
FLAG_DOCUMENTS_ONLY = 1
FLAG_ACCUM_SCORE = 2
OP_TERM, OP_AND, OP_OR, OP_PHRASE, OP_NOT, OP_OPT, OP_SOME = 0, 1, 2, 3, 4, 5, 6
SIM_BM25, SIM_TFIDF, SIM_TRIVIAL = 0, 1, 2


def tok(op, arg):
    return (op << 28) | (arg & 0x0FFFFFFF)


class ToTerm(C.Structure):
    _fields_ = [("documents", C.c_uint32), ("offset", C.c_uint32), ("size", C.c_uint32)]


class ToIndex(C.Structure):
    _fields_ = [
        ("bytes", C.POINTER(C.c_uint8)),
        ("len", C.c_size_t),
        ("terms", C.POINTER(ToTerm)),
        ("nterms", C.c_uint32),
        ("sumTermHits", C.c_uint64),
        ("totalTerms", C.c_uint32),
        ("sumTermsDocs", C.c_uint64),
        ("docsCnt", C.c_uint32),
        ("owns", C.c_int),
        ("masked", C.POINTER(C.c_uint32)),
        ("nmasked", C.c_size_t),
        ("codec", C.c_int),
        ("hits", C.POINTER(C.c_uint8)),
        ("hits_len", C.c_size_t),
        ("similarity", C.c_int),
    ]


class ToCorpus(C.Structure):
    _fields_ = [
        ("D", C.c_uint32),
        ("V", C.c_uint32),
        ("slots", C.c_uint32),
        ("ntokens", C.c_uint64),
        ("term_off", C.POINTER(C.c_uint64)),
        ("tok_doc", C.POINTER(C.c_uint32)),
        ("tok_pos", C.POINTER(C.c_uint16)),
    ]


class ToResult(C.Structure):
    _fields_ = [("docs", C.POINTER(C.c_uint32)), ("scores", C.POINTER(C.c_double)), ("n", C.c_size_t), ("cap", C.c_size_t)]


_lib = None


def build():
    """(Re)build the oracle shared object (and, when /root/reference exists, oracle/_ref/ref_driver)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    u8p, u16p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.to_varbyte_put32.restype = C.c_size_t
    L.to_varbyte_put32.argtypes = [u8p, C.c_uint32]
    L.to_varbyte_get32.restype = C.c_size_t
    L.to_varbyte_get32.argtypes = [u8p, u32p]
    L.to_corpus_generate.restype = C.POINTER(ToCorpus)
    L.to_corpus_generate.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]
    L.to_corpus_free.argtypes = [C.POINTER(ToCorpus)]
    L.to_gen_queries.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, u32p]
    L.to_google_encode.restype = C.POINTER(ToIndex)
    L.to_google_encode.argtypes = [C.POINTER(ToCorpus)]
    L.to_lucene_encode.restype = C.POINTER(ToIndex)
    L.to_lucene_encode.argtypes = [C.POINTER(ToCorpus)]
    L.to_ints_encode.restype = C.c_size_t
    L.to_ints_encode.argtypes = [C.c_void_p, C.c_void_p]
    L.to_ints_decode.restype = C.c_size_t
    L.to_ints_decode.argtypes = [C.c_void_p, C.c_void_p]
    L.to_index_wrap.restype = C.POINTER(ToIndex)
    L.to_index_wrap.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64]
    L.to_index_free.argtypes = [C.POINTER(ToIndex)]
    L.to_google_chunk_stats.restype = C.c_uint32
    L.to_google_chunk_stats.argtypes = [C.POINTER(ToIndex), C.c_uint32, u64p, u64p, u64p, u64p, u64p]
    L.to_pli_new.restype = C.c_void_p
    L.to_pli_new.argtypes = [C.POINTER(ToIndex), C.c_uint32]
    L.to_pli_free.argtypes = [C.c_void_p]
    for f in ("to_pli_next", "to_pli_current", "to_pli_freq"):
        getattr(L, f).restype = C.c_uint32
        getattr(L, f).argtypes = [C.c_void_p]
    L.to_pli_advance.restype = C.c_uint32
    L.to_pli_advance.argtypes = [C.c_void_p, C.c_uint32]
    L.to_pli_materialize_positions.restype = C.c_uint32
    L.to_pli_materialize_positions.argtypes = [C.c_void_p, u16p]
    L.to_pli_materialize_hits.restype = C.c_uint32
    L.to_pli_materialize_hits.argtypes = [C.c_void_p, u16p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)]
    L.to_decode_term.restype = C.c_uint32
    L.to_decode_term.argtypes = [C.POINTER(ToIndex), C.c_uint32, C.c_void_p, C.c_void_p]
    L.to_bm25_idf.restype = C.c_double
    L.to_bm25_idf.argtypes = [C.c_uint32, C.c_uint64]
    L.to_bm25_score.restype = C.c_float
    L.to_bm25_score.argtypes = [C.c_double, C.c_uint16]
    L.to_exec_query.restype = C.c_int
    L.to_exec_query.argtypes = [C.POINTER(ToIndex), C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(ToResult)]
    L.to_result_free.argtypes = [C.POINTER(ToResult)]
    L.to_topk.restype = C.c_uint32
    L.to_topk.argtypes = [C.POINTER(ToResult), C.c_uint32, C.c_void_p, C.c_void_p]
    L.to_fnv1a_docs.restype = C.c_uint64
    L.to_fnv1a_docs.argtypes = [C.c_void_p, C.c_size_t]
    L.to_splitmix64.restype = C.c_uint64
    L.to_splitmix64.argtypes = [u64p]
    _lib = L
    return L


def fnv1a_docs(docs):
    a = np.ascontiguousarray(docs, dtype=np.uint32)
    return int(lib().to_fnv1a_docs(a.ctypes.data, a.size))


def fnv1a_u32_stream(values):
    """FNV-1a over little-endian u32s (== ref_driver's fnv_u32 chaining)."""
    return fnv1a_docs(values)


def gen_queries(V, seed, nq, nterms):
    out = np.zeros((nq, nterms), dtype=np.uint32)
    lib().to_gen_queries(V, seed, nq, nterms, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


class Index:
    """Owns an oracle-side index (Google codec)."""

    def __init__(self, ptr, corpus=None):
        self.ptr = ptr
        self.corpus = corpus

    @classmethod
    def generate(cls, D, V, slots=10, seed=42, codec="google"):
        L = lib()
        c = L.to_corpus_generate(D, V, slots, seed)
        ix = L.to_lucene_encode(c) if codec == "lucene" else L.to_google_encode(c)
        return cls(ix, c)

    def hits(self):
        return np.ctypeslib.as_array(self.c.hits, shape=(self.c.hits_len,)).copy() if self.c.hits_len else np.zeros(0, np.uint8)

    @classmethod
    def wrap(cls, index_bytes, terms, docs_cnt, sum_terms_docs=0, sum_term_hits=0):
        b = np.ascontiguousarray(index_bytes, dtype=np.uint8)
        t = np.ascontiguousarray(terms, dtype=np.uint32).reshape(-1, 3)
        p = lib().to_index_wrap(b.ctypes.data, b.size, t.ctypes.data, t.shape[0], docs_cnt, sum_terms_docs, sum_term_hits)
        return cls(p)

    def __del__(self):
        try:
            if self.ptr:
                lib().to_index_free(self.ptr)
            if self.corpus:
                lib().to_corpus_free(self.corpus)
        except Exception:
            pass

    @property
    def c(self):
        return self.ptr.contents

    def bytes(self):
        return np.ctypeslib.as_array(self.c.bytes, shape=(self.c.len,)).copy()

    def terms(self):
        """(nterms, 3) u32: documents, offset, size — term_index_ctx."""
        a = np.ctypeslib.as_array(C.cast(self.c.terms, C.POINTER(C.c_uint32)), shape=(self.c.nterms, 3))
        return a.copy()

    def decode_term(self, t):
        n = int(self.c.terms[t].documents)
        docs = np.zeros(n, dtype=np.uint32)
        freqs = np.zeros(n, dtype=np.uint32)
        m = lib().to_decode_term(self.ptr, t, docs.ctypes.data, freqs.ctypes.data)
        assert m == n, (m, n)
        return docs, freqs

    def chunk_stats(self, t):
        v = [C.c_uint64() for _ in range(5)]
        blocks = lib().to_google_chunk_stats(self.ptr, t, *[C.byref(x) for x in v])
        return dict(blocks=blocks, hdr=v[0].value, docfreq=v[1].value, hits=v[2].value, skip=v[3].value, postings=v[4].value)

    def exec(self, prog, flags):
        """Run one postfix program; returns (docs u32[], scores f64[] or None)."""
        p = np.ascontiguousarray(prog, dtype=np.uint32)
        r = ToResult()
        rc = lib().to_exec_query(self.ptr, p.ctypes.data, p.size, flags, C.byref(r))
        if rc != 0:
            raise ValueError(f"to_exec_query rc={rc}")
        docs = np.ctypeslib.as_array(r.docs, shape=(r.n,)).copy() if r.n else np.zeros(0, np.uint32)
        scores = None
        if flags & FLAG_ACCUM_SCORE:
            scores = np.ctypeslib.as_array(r.scores, shape=(r.n,)).copy() if r.n else np.zeros(0, np.float64)
        lib().to_result_free(C.byref(r))
        return docs, scores

    def exec_rich(self, prog):
        """Default ("rich match") mode: returns (docs u32[], flat u32[]) — flat = per match: doc, nterms, then per matched term
        (ascending rank) rank, freq, pos[freq]; plus (terms_total, hits_total)."""

        class ToRich(C.Structure):
            _fields_ = [("docs", C.POINTER(C.c_uint32)), ("n", C.c_size_t), ("flat", C.POINTER(C.c_uint32)), ("nflat", C.c_size_t), ("capflat", C.c_size_t),
                        ("terms_total", C.c_uint64), ("hits_total", C.c_uint64)]

        p = np.ascontiguousarray(prog, dtype=np.uint32)
        r = ToRich()
        f = lib().to_exec_query_rich
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        rc = f(self.ptr, p.ctypes.data, p.size, C.byref(r))
        if rc != 0:
            raise ValueError(f"to_exec_query_rich rc={rc}")
        docs = np.ctypeslib.as_array(r.docs, shape=(r.n,)).copy() if r.n else np.zeros(0, np.uint32)
        flat = np.ctypeslib.as_array(r.flat, shape=(r.nflat,)).copy() if r.nflat else np.zeros(0, np.uint32)
        tt, ht = int(r.terms_total), int(r.hits_total)
        lib().to_rich_free.argtypes = [C.c_void_p]
        lib().to_rich_free(C.byref(r))
        return docs, flat, tt, ht

    def set_similarity(self, sim):
        """SIM_BM25 (default) / SIM_TFIDF / SIM_TRIVIAL: the scorer AccumulatedScoreScheme queries use (similarity.h)."""
        self.ptr.contents.similarity = int(sim)

    def set_masked(self, docids):
        d = np.ascontiguousarray(docids, dtype=np.uint32)
        f = lib().to_index_set_masked
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        f(self.ptr, d.ctypes.data, d.size)

    def exec_count(self, prog, flags):
        """Run one postfix program and return only the number of matches (no copy: the multi-threaded CPU-baseline leg of
        bench.py calls this from many threads; ctypes drops the GIL for the duration of the C call)."""
        p = np.ascontiguousarray(prog, dtype=np.uint32)
        r = ToResult()
        rc = lib().to_exec_query(self.ptr, p.ctypes.data, p.size, flags, C.byref(r))
        if rc != 0:
            raise ValueError(f"to_exec_query rc={rc}")
        n = int(r.n)
        lib().to_result_free(C.byref(r))
        return n

    def exec_batch_mt(self, progs, flags, nthreads, budget_s):
        """progs: [nq, proglen] u32.  One query per thread (C, pthreads) until the batch or the time budget runs out.
        Returns (queries done, matches, wall seconds)."""
        p = np.ascontiguousarray(progs, dtype=np.uint32)
        m, sec = C.c_uint64(), C.c_double()
        f = lib().to_exec_batch_mt
        f.restype = C.c_uint64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        n = f(self.ptr, p.ctypes.data, p.shape[1], p.shape[0], flags, nthreads, budget_s, C.byref(m), C.byref(sec))
        return int(n), int(m.value), float(sec.value)

    def topk(self, docs, scores, k):
        r = ToResult()
        d = np.ascontiguousarray(docs, dtype=np.uint32)
        s = np.ascontiguousarray(scores, dtype=np.float64)
        r.docs = d.ctypes.data_as(C.POINTER(C.c_uint32))
        r.scores = s.ctypes.data_as(C.POINTER(C.c_double))
        r.n = r.cap = d.size
        od = np.zeros(k, np.uint32)
        os_ = np.zeros(k, np.float32)
        m = lib().to_topk(C.byref(r), k, od.ctypes.data, os_.ctypes.data)
        return od[:m], os_[:m]


class PLI:
    def __init__(self, index, term):
        self.index = index
        self.p = lib().to_pli_new(index.ptr, term)

    def __del__(self):
        try:
            lib().to_pli_free(self.p)
        except Exception:
            pass

    def next(self):
        return lib().to_pli_next(self.p)

    def advance(self, t):
        return lib().to_pli_advance(self.p, t & 0xFFFFFFFF)

    def current(self):
        return lib().to_pli_current(self.p)

    def freq(self):
        return lib().to_pli_freq(self.p)

    def positions(self):
        buf = (C.c_uint16 * 65536)()
        n = lib().to_pli_materialize_positions(self.p, buf)
        return list(buf[:n])

    def hits(self):
        """(positions, payload lengths, payload words) of the current document — term_hit as materialize_hits fills it"""
        pos, ln, pl = (C.c_uint16 * 65536)(), (C.c_uint8 * 65536)(), (C.c_uint64 * 65536)()
        n = lib().to_pli_materialize_hits(self.p, pos, ln, pl)
        return list(pos[:n]), list(ln[:n]), list(pl[:n])


# ---- tiny query-text -> postfix program compiler for the query templates of SURVEY §8(d) ------------
def parse_query(text, some_min=1):
    """Supports: terms tN, juxtaposition = AND, OR, NOT, parentheses, "phrases", <optional>.  OR binds looser than AND
    (Trinity: queries.h operators; `a b OR c` is not used by the fixtures to avoid precedence ambiguity)."""
    toks = []
    i = 0
    while i < len(text):
        ch = text[i]
        if ch.isspace():
            i += 1
        elif ch in "()<>[],":
            toks.append(ch)
            i += 1
        elif ch == '"':
            j = text.index('"', i + 1)
            toks.append(("PHRASE", [int(w[1:]) for w in text[i + 1 : j].split()]))
            i = j + 1
        else:
            j = i
            while j < len(text) and not text[j].isspace() and text[j] not in '()"<>[],':
                j += 1
            w = text[i:j]
            toks.append(w if w in ("OR", "NOT") else ("TERM", int(w[1:])))
            i = j
    pos = [0]

    def peek():
        return toks[pos[0]] if pos[0] < len(toks) else None

    def primary():
        t = peek()
        pos[0] += 1
        if t == "(":
            r = expr_or()
            assert peek() == ")"
            pos[0] += 1
            return r
        if t == "[":  # [a, b, ...]: MatchSome (ast_parser::Flags::ParseMatchSomeExpr); the threshold is set on the node by the application
            kids = [expr_or()]
            while peek() == ",":
                pos[0] += 1
                kids.append(expr_or())
            assert peek() == "]"
            pos[0] += 1
            return sum(kids, []) + [tok(OP_SOME, (min(some_min, len(kids)) << 16) | len(kids))]
        if t == "<":  # <expr>: ConstTrueExpr (ast_parser::Flags::ParseConstTrueExpr) — optional under an AND
            r = expr_or()
            assert peek() == ">"
            pos[0] += 1
            return ("OPT", r)
        if t[0] == "TERM":
            return [tok(OP_TERM, t[1])]
        if t[0] == "PHRASE":
            if len(t[1]) == 1:
                return [tok(OP_TERM, t[1][0])]
            return [tok(OP_TERM, x) for x in t[1]] + [tok(OP_PHRASE, len(t[1]))]
        raise ValueError(t)

    def expr_and():
        parts = [primary()]
        while peek() is not None and peek() not in (")", ">", "]", ",", "OR", "NOT"):
            parts.append(primary())
        # juxtaposition is a left-associative binary AND in Trinity; an AND with a <...> operand becomes Optional(other, opt)
        # (exec.cpp:366-377), and a <...> that is not under an AND is just its expression (:434-441)
        req = [p_ for p_ in parts if not (isinstance(p_, tuple) and p_[0] == "OPT")]
        opts = [p_[1] for p_ in parts if isinstance(p_, tuple) and p_[0] == "OPT"]
        if not req:
            req, opts = [opts[0]], opts[1:]
        r = req[0] if len(req) == 1 else sum(req, []) + [tok(OP_AND, len(req))]
        for o in opts:
            r = r + o + [tok(OP_OPT, 2)]
        return r

    def expr_not():
        # `x y NOT z` == (x y) NOT z, left-associative (what Trinity's parser produced for the fixtures)
        r = expr_and()
        while peek() == "NOT":
            pos[0] += 1
            r = r + expr_and() + [tok(OP_NOT, 2)]
        return r

    def expr_or():
        parts = [expr_not()]
        while peek() == "OR":
            pos[0] += 1
            parts.append(expr_not())
        if len(parts) == 1:
            return parts[0]
        return sum(parts, []) + [tok(OP_OR, len(parts))]

    r = expr_or()
    assert pos[0] == len(toks), text
    return np.array(r, dtype=np.uint32)


def masked_docs(D, seed, permille):
    """The documents ref_driver's `filter <seed> <permille>` rules out (oracle/ref_driver.cpp HashFilter): d in 1..D with
    splitmix64(seed + d) % 1000 < permille."""
    x = (np.arange(1, D + 1, dtype=np.uint64) + np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    x = x ^ (x >> np.uint64(31))
    return (np.nonzero(x % np.uint64(1000) < np.uint64(permille))[0] + 1).astype(np.uint32)


def _plain_env():
    """The reference driver is not a sanitizer build: it runs without the preloaded runtime tests/test_oracle_sanitized.py puts in the environment."""
    return {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}


def run_ref_driver(D, V, slots, seed, commands):
    """Run the genuine reference (oracle/_ref/ref_driver) over the same corpus; returns parsed JSON lines."""
    import json

    out = subprocess.run([REF_DRIVER, str(D), str(V), str(slots), str(seed)], input="\n".join(commands) + "\n", capture_output=True, text=True, check=True, env=_plain_env())
    return [json.loads(l) for l in out.stdout.splitlines() if l.strip()]


def run_ref_driver_edge(commands):
    """The genuine reference over its EDGE corpus (oracle/ref_driver.cpp: payload-bearing hits, position-0 hits, a document with
    more than 65535 hits, positions up to MaxPosition - 1, repeated positions)."""
    import json

    out = subprocess.run([REF_DRIVER, "edge"], input="\n".join(commands) + "\n", capture_output=True, text=True, check=True, env=_plain_env())
    return [json.loads(l) for l in out.stdout.splitlines() if l.strip()]


def fnv1a_u32s(values, h=1469598103934665603):
    """FNV-1a(64) over little-endian u32 values (ref_driver's fnv_u32 chain)."""
    for v in values:
        v = int(v) & 0xFFFFFFFF
        for _ in range(4):
            h = ((h ^ (v & 0xFF)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
            v >>= 8
    return h


def program_from_exec_tree(tree):
    """Lower an exec_node tree as the reference's compile_query produces it (ref_driver `tree`, tests/golden/ref_trees.json) to the
    C-ABI's postfix program — node by node what exec.cpp:253-449 build_iterator does with it: matchallterms -> Conjuction of the
    run's terms, matchanyterms -> Disjunction, matchphrase -> Phrase, matchsome -> DisjunctionSome(min), logicalnot -> Filter,
    logicaland with a consttrueexpr side -> Optional(other side, expression) (exec.cpp:366-377), logicaland / logicalor otherwise."""
    op = tree["op"]

    def terms():
        return [tok(OP_TERM, t) for t in tree["t"]]

    if op == "term":
        return terms()
    if op == "allterms":
        return terms() + [tok(OP_AND, len(tree["t"]))]
    if op == "anyterms":
        return terms() + [tok(OP_OR, len(tree["t"]))]
    if op == "phrase":
        return terms() + ([tok(OP_PHRASE, len(tree["t"]))] if len(tree["t"]) > 1 else [])
    kids = tree.get("k", [])
    if op == "and":
        lhs, rhs = kids
        if lhs["op"] == "consttrueexpr" or rhs["op"] == "consttrueexpr":
            main, opt = (rhs, lhs) if lhs["op"] == "consttrueexpr" else (lhs, rhs)
            return program_from_exec_tree(main) + program_from_exec_tree(opt["k"][0]) + [tok(OP_OPT, 2)]
        return program_from_exec_tree(lhs) + program_from_exec_tree(rhs) + [tok(OP_AND, 2)]
    if op == "or":
        return sum((program_from_exec_tree(k) for k in kids), []) + [tok(OP_OR, len(kids))]
    if op == "not":
        return program_from_exec_tree(kids[0]) + program_from_exec_tree(kids[1]) + [tok(OP_NOT, 2)]
    if op == "some":
        return sum((program_from_exec_tree(k) for k in kids), []) + [tok(OP_SOME, (tree["min"] << 16) | len(kids))]
    if op in ("unaryand", "consttrueexpr"):  # (a lone expression: the documents of its operand)
        return program_from_exec_tree(kids[0])
    if op in ("allphrases", "anyphrases"):
        return sum((program_from_exec_tree(k) for k in kids), []) + [tok(OP_AND if op == "allphrases" else OP_OR, len(kids))]
    raise ValueError(f"exec_node {op} has no iterator")
