"""ctypes harness over the C-ABI (include/trinity_hip.h) and the host tools (csrc/host/synth.cpp).

No fallbacks: a missing libtrinity_hip.so, a missing GPU or a failing call raises TrinityError."""
import ctypes as C
import os

import numpy as np

from .build import LIB_HIP, LIB_HOST

# perf probes: an alternative build of the engine (build/libtrinity_hip_*.so: -DTRI_PROF, other launch bounds ...) for this process.
# The harness only — the library itself reads nothing from the environment.
LIB_HIP = os.environ.get("TRINITY_HIP_LIB") or LIB_HIP
LIB_HOST = os.environ.get("TRINITY_HOST_LIB") or LIB_HOST  # (the host tools under ThreadSanitizer: tests/test_planner_tsan.py)

OP_TERM, OP_AND, OP_OR, OP_PHRASE, OP_NOT, OP_OPT, OP_SOME = 0, 1, 2, 3, 4, 5, 6  # OP_SOME: arg = (min << 16) | children (matchsome)
FLAG_DOCUMENTS_ONLY, FLAG_ACCUMULATED_SCORE, FLAG_MATCHED_TERMS, FLAG_HIT_PAYLOADS = 1, 2, 4, 8
CODEC_GOOGLE, CODEC_LUCENE = 1, 2
FNV_EMPTY = 1469598103934665603


class TrinityError(RuntimeError):
    pass


def tok(op, arg):
    return (op << 28) | (arg & 0x0FFFFFFF)


class TriTerm(C.Structure):
    _fields_ = [("documents", C.c_uint32), ("offset", C.c_uint32), ("size", C.c_uint32)]


class TriQuery(C.Structure):
    _fields_ = [("prog_off", C.c_uint32), ("prog_len", C.c_uint32)]


class TriIndexInfo(C.Structure):
    _fields_ = [
        ("index_bytes", C.c_uint64),
        ("directory_bytes", C.c_uint64),
        ("blocks", C.c_uint64),
        ("postings", C.c_uint64),
        ("doc_bytes", C.c_uint64),
        ("hit_bytes", C.c_uint64),
        ("nterms", C.c_uint32),
        ("docs_cnt", C.c_uint32),
    ]


class TriBatchInfo(C.Structure):
    _fields_ = [
        ("nqueries", C.c_uint64),
        ("algorithmic_bytes", C.c_uint64),
        ("matches", C.c_uint64),
        ("out_capacity", C.c_uint64),
        ("last_run_ms", C.c_float),
        ("launches", C.c_uint32),
        ("dense_ms", C.c_float),
        ("cand_ms", C.c_float),
        ("dense_algorithmic_bytes", C.c_uint64),
        ("cand_algorithmic_bytes", C.c_uint64),
        ("dense_queries", C.c_uint64),
        ("cand_queries", C.c_uint64),
        ("fused_ms", C.c_float),
        ("rest_ms", C.c_float),
        ("fused_algorithmic_bytes", C.c_uint64),
        ("fused_queries", C.c_uint64),
        ("cand_needed_bytes", C.c_uint64),
        ("phrase_ms", C.c_float),
        ("tree_ms", C.c_float),
        ("phrase_algorithmic_bytes", C.c_uint64),
        ("phrase_queries", C.c_uint64),
        ("term_planes_ms", C.c_float),
        ("planes_ms", C.c_float),
        ("planes_algorithmic_bytes", C.c_uint64),
        ("planes_queries", C.c_uint64),
        ("plane_terms", C.c_uint64),
        ("plane_bytes", C.c_uint64),
        ("term_planes_decoded_bytes", C.c_uint64),
        ("unsupported_queries", C.c_uint64),
        ("create_ms", C.c_float),
        ("create_plan_ms", C.c_float),
        ("bound_bytes", C.c_uint64),
        ("dense_bound_bytes", C.c_uint64),
        ("cand_bound_bytes", C.c_uint64),
        ("fused_bound_bytes", C.c_uint64),
        ("planes_bound_bytes", C.c_uint64),
        ("phrase_bound_bytes", C.c_uint64),
        ("pset_ms", C.c_float),
        ("probe_ms", C.c_float),
        ("pset_queries", C.c_uint64),
        ("pset_algorithmic_bytes", C.c_uint64),
        ("pset_bound_bytes", C.c_uint64),
        ("probe_queries", C.c_uint64),
        ("probe_algorithmic_bytes", C.c_uint64),
        ("probe_bound_bytes", C.c_uint64),
        ("tree_queries", C.c_uint64),
        ("tree_scratch_bytes", C.c_uint64),
        ("bitmap_queries", C.c_uint64),
    ]


# every symbol include/trinity_hip.h declares (tests/test_abi.py checks the library exports all of them)
ABI_SYMBOLS = [
    "tri_last_error", "tri_abi_version", "tri_dev_open", "tri_dev_close", "tri_dev_sync", "tri_dev_stream", "tri_dev_set_option", "tri_dev_get_option", "tri_dev_memory",
    "tri_index_upload", "tri_index_destroy", "tri_index_get_info", "tri_index_term_docbytes", "tri_index_set_masked", "tri_decode_terms",
    "tri_batch_create", "tri_batch_query_status", "tri_batch_destroy", "tri_batch_run", "tri_batch_sync", "tri_batch_get_info",
    "tri_batch_match_counts", "tri_batch_docset", "tri_batch_docset_bitmap", "tri_batch_docsets", "tri_batch_docsets_mixed", "tri_batch_scores", "tri_batch_query_terms", "tri_batch_matched_terms", "tri_batch_matched_payloads", "tri_batch_topk", "tri_batch_topk_device", "tri_batch_counts_device", "tri_batch_docset_hashes",
    "tri_cbatch_create", "tri_cbatch_destroy", "tri_cbatch_query_status", "tri_cbatch_run", "tri_cbatch_sync", "tri_cbatch_match_counts", "tri_cbatch_topk", "tri_cbatch_docset", "tri_encode_google", "tri_encode_google_payloads", "tri_commit_google", "tri_commit_lucene", "tri_merge_google", "tri_merge_lucene", "tri_encode_lucene",
    "tri_comm_unique_id", "tri_comm_create", "tri_comm_create_custom", "tri_comm_destroy", "tri_gather_results",
]  # fmt: skip

_hip = None
_host = None


def hip_lib():
    """Load libtrinity_hip.so; raises (never falls back) when it is missing."""
    global _hip
    if _hip is not None:
        return _hip
    if not os.path.exists(LIB_HIP):
        raise TrinityError(f"{LIB_HIP} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` — there is no CPU fallback")
    L = C.CDLL(LIB_HIP)
    vp, u32p, u64p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.tri_last_error.restype = C.c_char_p
    L.tri_abi_version.restype = C.c_int
    L.tri_dev_open.argtypes = [C.c_int, C.POINTER(vp)]
    L.tri_dev_close.argtypes = [vp]
    L.tri_dev_sync.argtypes = [vp]
    L.tri_dev_stream.restype = vp
    L.tri_dev_stream.argtypes = [vp]
    L.tri_dev_set_option.argtypes = [vp, C.c_char_p, C.c_uint64]
    L.tri_dev_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint64)]
    L.tri_dev_memory.argtypes = [vp, C.POINTER(C.c_uint64 * 5)]
    L.tri_index_upload.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int, vp, C.c_size_t, C.c_uint32, C.POINTER(vp)]
    L.tri_index_destroy.argtypes = [vp]
    L.tri_index_get_info.argtypes = [vp, C.POINTER(TriIndexInfo)]
    L.tri_index_term_docbytes.argtypes = [vp, vp, C.c_size_t, vp]
    L.tri_index_set_masked.argtypes = [vp, vp, C.c_size_t]
    L.tri_decode_terms.argtypes = [vp, vp, C.c_size_t, vp, vp, vp]
    L.tri_batch_create.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp)]
    L.tri_batch_query_status.argtypes = [vp, vp]
    L.tri_batch_destroy.argtypes = [vp]
    L.tri_batch_run.argtypes = [vp]
    L.tri_batch_sync.argtypes = [vp]
    L.tri_batch_get_info.argtypes = [vp, C.POINTER(TriBatchInfo)]
    L.tri_batch_query_terms.argtypes = [vp, C.c_size_t, vp, C.POINTER(C.c_uint32)]
    L.tri_batch_matched_terms.argtypes = [vp, C.c_size_t, vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.tri_batch_matched_payloads.argtypes = [vp, C.c_size_t, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.tri_batch_match_counts.argtypes = [vp, vp]
    L.tri_batch_docset.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.tri_batch_docsets.argtypes = [vp, vp, C.c_size_t, vp]
    L.tri_batch_docsets_mixed.argtypes = [vp, vp, C.c_size_t, vp, vp]
    L.tri_batch_docset_bitmap.argtypes = [vp, C.c_size_t, C.POINTER(C.c_int), vp, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_size_t)]
    L.tri_batch_scores.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.tri_batch_topk.argtypes = [vp, vp, vp, vp]
    L.tri_batch_topk_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.tri_batch_counts_device.argtypes = [vp, C.POINTER(vp)]
    L.tri_batch_docset_hashes.argtypes = [vp, vp]
    L.tri_cbatch_create.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.tri_cbatch_destroy.argtypes = [vp]
    L.tri_cbatch_query_status.argtypes = [vp, vp]
    L.tri_cbatch_run.argtypes = [vp]
    L.tri_cbatch_sync.argtypes = [vp]
    L.tri_cbatch_match_counts.argtypes = [vp, vp]
    L.tri_cbatch_topk.argtypes = [vp, vp, vp, vp]
    L.tri_cbatch_docset.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.tri_encode_google.argtypes = [vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t), vp]
    L.tri_encode_google_payloads.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t), vp]
    L.tri_encode_lucene.argtypes = [vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t), vp, C.c_size_t, C.POINTER(C.c_size_t), vp]
    L.tri_commit_lucene.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t), vp, C.c_size_t, C.POINTER(C.c_size_t), vp, vp, C.c_size_t,
                                    C.POINTER(C.c_size_t), vp]
    L.tri_merge_google.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t), vp, vp]
    L.tri_commit_google.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t), vp, vp, C.c_size_t, C.POINTER(C.c_size_t), vp]
    L.tri_comm_unique_id.argtypes = [vp]
    L.tri_comm_create.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.tri_comm_create_custom.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.POINTER(vp)]
    L.tri_comm_destroy.argtypes = [vp]
    L.tri_gather_results.argtypes = [vp, vp, vp, vp, vp, vp]
    _hip = L
    return L


def host_lib():
    global _host
    if _host is not None:
        return _host
    if not os.path.exists(LIB_HOST):
        raise TrinityError(f"{LIB_HOST} is missing: run __graft_entry__.build()")
    L = C.CDLL(LIB_HOST)
    L.tri_synth_segment_build.restype = C.c_void_p
    L.tri_synth_segment_build.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]
    L.tri_synth_segment_build_codec.restype = C.c_void_p
    L.tri_synth_segment_build_codec.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int]
    L.tri_synth_segment_hits.restype = C.POINTER(C.c_uint8)
    L.tri_synth_segment_hits.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.tri_synth_segment_free.argtypes = [C.c_void_p]
    L.tri_host_encode_google.restype = C.c_longlong
    L.tri_host_encode_google.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    L.tri_host_encode_google_payloads.restype = C.c_longlong
    L.tri_host_encode_google_payloads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    L.tri_synth_segment_index.restype = C.POINTER(C.c_uint8)
    L.tri_synth_segment_index.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.tri_synth_segment_terms.restype = C.POINTER(C.c_uint32)
    L.tri_synth_segment_terms.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.tri_synth_segment_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.tri_synth_queries.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    _host = L
    return L


def _check(rc):
    if rc != 0:
        raise TrinityError(f"rc={rc}: {hip_lib().tri_last_error().decode()}")


def gen_queries(V, seed, nq, nterms):
    out = np.zeros((nq, nterms), dtype=np.uint32)
    host_lib().tri_synth_queries(V, seed, nq, nterms, out.ctypes.data)
    return out


def gen_phrase_queries(D, V, slots, corpus_seed, seed, nq, nterms):
    """cfg4 phrases: even rows are consecutive tokens of a random document (>= 1 match), odd rows random Zipf terms."""
    out = np.zeros((nq, nterms), dtype=np.uint32)
    L = host_lib()
    L.tri_synth_phrase_queries.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    L.tri_synth_phrase_queries.restype = None
    L.tri_synth_phrase_queries(D, V, slots, corpus_seed, seed, nq, nterms, out.ctypes.data)
    return out


def flatten(programs):
    """(flat u32 program, (nq, 2) u32 tri_query table of (offset, length)) of a list of postfix programs — laid out with numpy (a ctypes
    struct array costs a microsecond per field store)."""
    progs = [np.ascontiguousarray(p, dtype=np.uint32) for p in programs]
    flat = np.concatenate(progs) if progs else np.zeros(0, np.uint32)
    q = np.zeros((max(1, len(progs)), 2), dtype=np.uint32)
    if progs:
        q[:, 1] = np.fromiter((p.size for p in progs), dtype=np.uint32, count=len(progs))
        q[1:, 0] = np.cumsum(q[:-1, 1], dtype=np.uint64).astype(np.uint32)
    return flat, q


class Segment:
    """A synthetic GOOGLE-codec segment built on the host (csrc/host/synth.cpp): raw `index` bytes + term table."""

    def __init__(self, D, V, slots=10, seed=42, codec=CODEC_GOOGLE):
        L = host_lib()
        # codec 3: the LUCENE-shaped container with FastPFor<4> payload words (what the reference's own build writes; csrc/fastpfor128.hpp) —
        # to everything downstream it is a LUCENE segment (codec 2)
        self.D, self.V, self.slots, self.seed, self.codec, self.payload = D, V, slots, seed, min(codec, CODEC_LUCENE), "fastpfor" if codec == 3 else "native"
        self.h = L.tri_synth_segment_build_codec(D, V, slots, seed, codec)
        if not self.h:
            raise TrinityError("segment build failed (index larger than 4 GiB?)")
        n = C.c_uint64()
        p = L.tri_synth_segment_index(self.h, C.byref(n))
        self.index = np.ctypeslib.as_array(p, shape=(n.value,))
        hp = L.tri_synth_segment_hits(self.h, C.byref(n))
        self.hits = np.ctypeslib.as_array(hp, shape=(n.value,)) if n.value else np.zeros(0, np.uint8)
        nt = C.c_uint32()
        tp = L.tri_synth_segment_terms(self.h, C.byref(nt))
        self.terms = np.ctypeslib.as_array(tp, shape=(nt.value, 3))
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32()
        L.tri_synth_segment_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        self.sum_terms_docs, self.sum_term_hits, self.total_terms, self.docs_cnt = a.value, b.value, c.value, d.value

    def __del__(self):
        try:
            if self.h:
                host_lib().tri_synth_segment_free(self.h)
                self.h = None
        except Exception:
            pass


class Device:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        _check(hip_lib().tri_dev_open(device, C.byref(self.h)))

    def sync(self):
        _check(hip_lib().tri_dev_sync(self.h))

    def set_option(self, name, value):
        """Planner / launch option (include/trinity_hip.h tri_dev_set_option); applies to batches created afterwards."""
        _check(hip_lib().tri_dev_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_uint64()
        _check(hip_lib().tri_dev_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def memory(self):
        """tri_dev_memory: the handle's buffer pool (in use / idle), its idle pinned blocks, and the device's free / total bytes."""
        v = (C.c_uint64 * 5)()
        _check(hip_lib().tri_dev_memory(self.h, C.byref(v)))
        return dict(zip(("pool_in_use_bytes", "pool_idle_bytes", "pinned_idle_bytes", "device_free_bytes", "device_total_bytes"), (int(x) for x in v)))

    def encode_google(self, docs, freqs, positions, term_first, payload_lens=None, payloads=None):
        """The write side on the device (tri_encode_google / tri_encode_google_payloads): postings of len(term_first) - 1 terms ->
        (index bytes u8[], term table u32[n, 3]).  payload_lens / payloads: per hit, its payload's length (0 .. 8) and bytes (u64, first byte low)."""
        d = np.ascontiguousarray(docs, dtype=np.uint32)
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        p = np.ascontiguousarray(positions, dtype=np.uint16)
        tf = np.ascontiguousarray(term_first, dtype=np.uint64)
        n = tf.size - 1
        terms = np.zeros((max(n, 1), 3), dtype=np.uint32)
        ln = C.c_size_t()
        L = hip_lib()
        if payload_lens is None:
            call = lambda out, cap: L.tri_encode_google(self.h, d.ctypes.data, f.ctypes.data, p.ctypes.data, p.size, tf.ctypes.data, n, out, cap, C.byref(ln), terms.ctypes.data)
        else:
            pl = np.ascontiguousarray(payload_lens, dtype=np.uint8)
            pv = np.ascontiguousarray(payloads, dtype=np.uint64)
            assert pl.size == p.size == pv.size
            call = lambda out, cap: L.tri_encode_google_payloads(self.h, d.ctypes.data, f.ctypes.data, p.ctypes.data, pl.ctypes.data, pv.ctypes.data, p.size, tf.ctypes.data, n, out,
                                                                  cap, C.byref(ln), terms.ctypes.data)  # fmt: skip
        _check(call(None, 0))
        out = np.zeros(max(1, ln.value), dtype=np.uint8)
        _check(call(out.ctypes.data, out.size))
        return out[: ln.value], terms[:n]

    def commit_google(self, term_ids, doc_ids, freqs, positions, payload_lens=None, payloads=None):
        """SegmentIndexSession::commit on the device (tri_commit_google): a session's postings in insertion order -> (index bytes u8[], the committed
        termIDs u32[n] in commit order, their term table u32[n, 3], stats {docs_cnt, sum_terms_docs, sum_term_hits, total_terms})."""
        t = np.ascontiguousarray(term_ids, dtype=np.uint32)
        d = np.ascontiguousarray(doc_ids, dtype=np.uint32)
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        p = np.ascontiguousarray(positions, dtype=np.uint16)
        pl = None if payload_lens is None else np.ascontiguousarray(payload_lens, dtype=np.uint8)
        pv = None if payload_lens is None else np.ascontiguousarray(payloads, dtype=np.uint64)
        assert t.size == d.size == f.size and (pl is None or pl.size == p.size == pv.size)
        ln, nt = C.c_size_t(), C.c_size_t()
        stats = np.zeros(4, dtype=np.uint64)
        L = hip_lib()

        def call(out, cap, tids, terms, tcap):
            return L.tri_commit_google(self.h, t.ctypes.data, d.ctypes.data, f.ctypes.data, p.ctypes.data, None if pl is None else pl.ctypes.data,
                                       None if pv is None else pv.ctypes.data, t.size, p.size, out, cap, C.byref(ln), tids, terms, tcap, C.byref(nt), stats.ctypes.data)  # fmt: skip

        _check(call(None, 0, None, None, 0))
        out = np.zeros(max(1, ln.value), dtype=np.uint8)
        tids = np.zeros(max(1, nt.value), dtype=np.uint32)
        terms = np.zeros((max(1, nt.value), 3), dtype=np.uint32)
        _check(call(out.ctypes.data, out.size, tids.ctypes.data, terms.ctypes.data, nt.value))
        return out[: ln.value], tids[: nt.value], terms[: nt.value], dict(zip(("docs_cnt", "sum_terms_docs", "sum_term_hits", "total_terms"), (int(x) for x in stats)))

    def commit_lucene(self, term_ids, doc_ids, freqs, positions):
        """tri_commit_lucene: a session's postings in insertion order -> (index bytes, hits.data bytes, committed termIDs, term table, stats)."""
        t = np.ascontiguousarray(term_ids, dtype=np.uint32)
        d = np.ascontiguousarray(doc_ids, dtype=np.uint32)
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        p = np.ascontiguousarray(positions, dtype=np.uint16)
        il, hl, nt = C.c_size_t(), C.c_size_t(), C.c_size_t()
        stats = np.zeros(4, dtype=np.uint64)
        L = hip_lib()

        def call(io, ic, ho, hc, tids, terms, tcap):
            return L.tri_commit_lucene(self.h, t.ctypes.data, d.ctypes.data, f.ctypes.data, p.ctypes.data, t.size, p.size, io, ic, C.byref(il), ho, hc, C.byref(hl), tids, terms, tcap,
                                       C.byref(nt), stats.ctypes.data)  # fmt: skip

        _check(call(None, 0, None, 0, None, None, 0))
        io, ho = np.zeros(max(1, il.value), dtype=np.uint8), np.zeros(max(1, hl.value), dtype=np.uint8)
        tids, terms = np.zeros(max(1, nt.value), dtype=np.uint32), np.zeros((max(1, nt.value), 3), dtype=np.uint32)
        _check(call(io.ctypes.data, io.size, ho.ctypes.data, ho.size, tids.ctypes.data, terms.ctypes.data, nt.value))
        return io[: il.value], ho[: hl.value], tids[: nt.value], terms[: nt.value], dict(zip(("docs_cnt", "sum_terms_docs", "sum_term_hits", "total_terms"), (int(x) for x in stats)))

    def encode_lucene(self, docs, freqs, positions, term_first):
        """The Lucene-shaped codec's encoder on the device (tri_encode_lucene, PFOR128 payload): -> (index bytes, hits.data bytes, term table u32[n, 3])."""
        d = np.ascontiguousarray(docs, dtype=np.uint32)
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        p = np.ascontiguousarray(positions, dtype=np.uint16)
        tf = np.ascontiguousarray(term_first, dtype=np.uint64)
        n = tf.size - 1
        terms = np.zeros((max(n, 1), 3), dtype=np.uint32)
        il, hl = C.c_size_t(), C.c_size_t()
        L = hip_lib()
        call = lambda io, ic, ho, hc: L.tri_encode_lucene(self.h, d.ctypes.data, f.ctypes.data, p.ctypes.data, p.size, tf.ctypes.data, n, io, ic, C.byref(il), ho, hc, C.byref(hl), terms.ctypes.data)
        _check(call(None, 0, None, 0))
        io, ho = np.zeros(max(1, il.value), dtype=np.uint8), np.zeros(max(1, hl.value), dtype=np.uint8)
        _check(call(io.ctypes.data, io.size, ho.ctypes.data, ho.size))
        return io[: il.value], ho[: hl.value], terms[:n]

    def merge_google(self, parts, part_terms):
        """Codecs::Google::IndexSession::merge for a whole dictionary (tri_merge_google): parts = uploaded google_codec Index objects, most recent first
        (each with its masked documents set); part_terms u32[nterms, nparts] = the output term's index in each part (0xffffffff: absent) ->
        (index bytes, term table u32[nterms, 3], stats)."""
        pt = np.ascontiguousarray(part_terms, dtype=np.uint32).reshape(-1, len(parts))
        hs = (C.c_void_p * len(parts))(*[p.h for p in parts])
        terms = np.zeros((max(1, pt.shape[0]), 3), dtype=np.uint32)
        ln = C.c_size_t()
        stats = np.zeros(4, dtype=np.uint64)
        L = hip_lib()
        call = lambda out, cap: L.tri_merge_google(self.h, hs, len(parts), pt.ctypes.data, pt.shape[0], out, cap, C.byref(ln), terms.ctypes.data, stats.ctypes.data)
        _check(call(None, 0))
        out = np.zeros(max(1, ln.value), dtype=np.uint8)
        _check(call(out.ctypes.data, out.size))
        return out[: ln.value], terms[: pt.shape[0]], dict(zip(("docs_cnt", "sum_terms_docs", "sum_term_hits", "total_terms"), (int(x) for x in stats)))

    def merge_lucene(self, parts, part_terms):
        """Codecs::Lucene::IndexSession::merge for a whole dictionary (tri_merge_lucene): parts = lucene_codec Index objects uploaded with their hits.data, most recent
        first -> (index bytes, hits.data bytes, term table u32[nterms, 3], stats)."""
        pt = np.ascontiguousarray(part_terms, dtype=np.uint32).reshape(-1, len(parts))
        hs = (C.c_void_p * len(parts))(*[p.h for p in parts])
        terms = np.zeros((max(1, pt.shape[0]), 3), dtype=np.uint32)
        ln, hl = C.c_size_t(), C.c_size_t()
        stats = np.zeros(4, dtype=np.uint64)
        L = hip_lib()
        L.tri_merge_lucene.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                       C.c_void_p, C.c_void_p]  # fmt: skip
        call = lambda out, cap, hout, hcap: L.tri_merge_lucene(self.h, hs, len(parts), pt.ctypes.data, pt.shape[0], out, cap, C.byref(ln), hout, hcap, C.byref(hl), terms.ctypes.data, stats.ctypes.data)
        _check(call(None, 0, None, 0))
        out, hout = np.zeros(max(1, ln.value), dtype=np.uint8), np.zeros(max(1, hl.value), dtype=np.uint8)
        _check(call(out.ctypes.data, out.size, hout.ctypes.data, hout.size))
        return out[: ln.value], hout[: hl.value], terms[: pt.shape[0]], dict(zip(("docs_cnt", "sum_terms_docs", "sum_term_hits", "total_terms"), (int(x) for x in stats)))

    def close(self):
        if self.h:
            hip_lib().tri_dev_close(self.h)
            self.h = C.c_void_p()


def host_encode_google(docs, freqs, positions, term_first, payload_lens=None, payloads=None):
    """The HOST encoder (csrc/host/google_encoder.hpp: byte-identical to the reference's) over the same arguments: tests' checker of
    Device.encode_google."""
    d = np.ascontiguousarray(docs, dtype=np.uint32)
    f = np.ascontiguousarray(freqs, dtype=np.uint32)
    p = np.ascontiguousarray(positions, dtype=np.uint16)
    tf = np.ascontiguousarray(term_first, dtype=np.uint64)
    pl = None if payload_lens is None else np.ascontiguousarray(payload_lens, dtype=np.uint8)
    pv = None if payload_lens is None else np.ascontiguousarray(payloads, dtype=np.uint64)
    n = tf.size - 1
    terms = np.zeros((max(n, 1), 3), dtype=np.uint32)
    out = np.zeros(int(d.size * 10 + p.size * 13 + 16 * (n + 1) + 64), dtype=np.uint8)
    L = host_lib()
    ln = L.tri_host_encode_google_payloads(d.ctypes.data, f.ctypes.data, p.ctypes.data, None if pl is None else pl.ctypes.data, None if pv is None else pv.ctypes.data,
                                           tf.ctypes.data, n, out.ctypes.data, out.size, terms.ctypes.data)  # fmt: skip
    if ln < 0:
        raise TrinityError("host encoder failed")
    return out[:ln], terms[:n]


class Index:
    def __init__(self, dev, index_bytes, terms, docs_cnt, codec=CODEC_GOOGLE, hits=None):
        self.dev = dev
        b = np.ascontiguousarray(index_bytes, dtype=np.uint8)
        t = np.ascontiguousarray(terms, dtype=np.uint32).reshape(-1, 3)
        h = np.ascontiguousarray(hits, dtype=np.uint8) if hits is not None and len(hits) else None
        self.nterms = t.shape[0]
        self.h = C.c_void_p()
        _check(hip_lib().tri_index_upload(dev.h, b.ctypes.data, b.size, h.ctypes.data if h is not None else None, h.size if h is not None else 0, codec, t.ctypes.data, t.shape[0], docs_cnt, C.byref(self.h)))

    @classmethod
    def from_segment(cls, dev, seg):
        return cls(dev, seg.index, seg.terms, seg.docs_cnt, codec=getattr(seg, "codec", CODEC_GOOGLE), hits=getattr(seg, "hits", None))

    def info(self):
        i = TriIndexInfo()
        _check(hip_lib().tri_index_get_info(self.h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in TriIndexInfo._fields_}

    def set_masked(self, docids):
        """Replace the segment's masked-document set (masked_documents_registry::test, docidupdates.h:90-119)."""
        d = np.ascontiguousarray(docids, dtype=np.uint32)
        _check(hip_lib().tri_index_set_masked(self.h, d.ctypes.data if d.size else None, d.size))

    def term_docbytes(self, terms):
        t = np.ascontiguousarray(terms, dtype=np.uint32)
        out = np.zeros(t.size, dtype=np.uint64)
        _check(hip_lib().tri_index_term_docbytes(self.h, t.ctypes.data, t.size, out.ctypes.data))
        return out

    def decode_terms(self, terms, df, want_freqs=True):
        """GPU decode of whole postings lists; df = documents per requested term (sizes the host buffers)."""
        t = np.ascontiguousarray(terms, dtype=np.uint32)
        tot = int(np.sum(df))
        docs = np.zeros(tot, dtype=np.uint32)
        freqs = np.zeros(tot, dtype=np.uint32) if want_freqs else None
        offs = np.zeros(t.size + 1, dtype=np.uint64)
        _check(hip_lib().tri_decode_terms(self.h, t.ctypes.data, t.size, docs.ctypes.data, freqs.ctypes.data if want_freqs else None, offs.ctypes.data))
        assert int(offs[-1]) == tot, (int(offs[-1]), tot)
        return docs, freqs, offs

    def close(self):
        if self.h:
            hip_lib().tri_index_destroy(self.h)
            self.h = C.c_void_p()


class Batch:
    """A compiled batch of postfix query programs."""

    def __init__(self, index, programs, flags, topk=0, similarity=0, allow_unsupported=False, flat=None):
        """programs: a list of postfix programs (u32 arrays) — or flat = flatten(programs), the (program, tri_query table) pair the C-ABI
        takes, prepared once by a caller that compiles the same queries again and again (bench.py's end-to-end loop).
        A query whose shape the planner does not lower is left out of the batch with a per-query status (it reports no matches): unless
        the caller says it handles that (allow_unsupported=True, then query_status()), such a batch raises instead of answering silently."""
        self.index = index
        self.nq = len(programs) if flat is None else flat[1].shape[0]
        flat, q = flatten(programs) if flat is None else flat
        self.flags, self.topk = flags, topk
        self.h = C.c_void_p()
        _check(hip_lib().tri_batch_create(index.h, flat.ctypes.data, flat.size, q.ctypes.data, self.nq, None, flags, topk, similarity, C.byref(self.h)))
        if not allow_unsupported:
            n = self.info()["unsupported_queries"]
            if n:
                msg = hip_lib().tri_last_error().decode()
                self.close()
                raise TrinityError(f"{n} queries of the batch have a shape the planner does not lower (allow_unsupported=True + query_status() to run the rest): {msg}")

    @classmethod
    def conjunctions(cls, index, term_rows, flags=FLAG_DOCUMENTS_ONLY, topk=0):
        rows = np.asarray(term_rows, dtype=np.uint32)
        k = rows.shape[1]
        progs = np.empty((rows.shape[0], k + 1), dtype=np.uint32)
        progs[:, :k] = rows  # TERM tokens: op 0 => the raw term id
        progs[:, k] = tok(OP_AND, k)
        return cls(index, list(progs), flags, topk)

    def query_status(self):
        """Per query: 0, or the status (TRI_ERR_UNSUPPORTED = -3) with which the planner left it out of the batch."""
        out = np.zeros(self.nq, dtype=np.int32)
        _check(hip_lib().tri_batch_query_status(self.h, out.ctypes.data))
        return out

    def run(self):
        _check(hip_lib().tri_batch_run(self.h))

    def sync(self):
        _check(hip_lib().tri_batch_sync(self.h))

    def info(self):
        i = TriBatchInfo()
        _check(hip_lib().tri_batch_get_info(self.h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in TriBatchInfo._fields_}

    def matched_terms(self, q, n):
        """FLAG_MATCHED_TERMS batches: (terms u32[nt], present u32[n], freq u16[n, nt], positions u16[npos]) for the n matches of
        query q (ascending docID, as docset(q, n) returns them); positions are match-major then term-minor."""
        L = hip_lib()
        terms = np.zeros(16, dtype=np.uint32)
        nt = C.c_uint32()
        _check(L.tri_batch_query_terms(self.h, q, terms.ctypes.data, C.byref(nt)))
        nt = nt.value
        npos = C.c_size_t()
        _check(L.tri_batch_matched_terms(self.h, q, None, None, None, 0, C.byref(npos)))
        present = np.zeros(n, dtype=np.uint32)
        freq = np.zeros((n, max(nt, 1)), dtype=np.uint16)
        pos = np.zeros(max(1, npos.value), dtype=np.uint16)
        _check(L.tri_batch_matched_terms(self.h, q, present.ctypes.data, freq.ctypes.data, pos.ctypes.data, pos.size, C.byref(npos)))
        return terms[:nt], present, freq[:, :nt], pos[: npos.value]

    def counts(self):
        out = np.zeros(self.nq, dtype=np.uint64)
        _check(hip_lib().tri_batch_match_counts(self.h, out.ctypes.data))
        return out

    def docset(self, q, n=None):
        if n is None:
            n = int(self.counts()[q])
        out = np.zeros(n, dtype=np.uint32)
        got = C.c_size_t()
        _check(hip_lib().tri_batch_docset(self.h, q, out.ctypes.data, n, C.byref(got)))
        return out[: got.value]

    def docsets(self, out=None):
        """Every query's docID set in one call (tri_batch_docsets): (flat u32[], offsets u64[nq + 1]) — query q's ascending docIDs are
        flat[offsets[q]:offsets[q + 1]].  `out`: a caller's u32 buffer (numpy array or anything with ctypes.data / data_ptr(): pinned memory makes
        the one device-to-host copy run at PCIe's rate); None: a fresh array of the exact size."""
        L = hip_lib()
        offs = np.zeros(self.nq + 1, dtype=np.uint64)
        _check(L.tri_batch_docsets(self.h, None, 0, offs.ctypes.data))
        total = int(offs[-1])
        if out is None:
            out = np.zeros(max(1, total), dtype=np.uint32)
        ptr, cap = (out.data_ptr(), out.numel()) if hasattr(out, "data_ptr") else (out.ctypes.data, out.size)
        _check(L.tri_batch_docsets(self.h, ptr, cap, offs.ctypes.data))
        return out, offs

    def docsets_mixed(self, out=None):
        """Every query's docID set in the form the engine holds it (tri_batch_docsets_mixed): (flat u32[], offsets u64[nq + 1], forms u32[nq]) — forms[q] = 0:
        flat[offsets[q]:offsets[q + 1]] = query q's ascending docIDs; 1: the words of a bitmap over its docID range (bit j of word i = document 32 i + j)."""
        L = hip_lib()
        offs = np.zeros(self.nq + 1, dtype=np.uint64)
        forms = np.zeros(max(1, self.nq), dtype=np.uint32)
        _check(L.tri_batch_docsets_mixed(self.h, None, 0, offs.ctypes.data, forms.ctypes.data))
        total = int(offs[-1])
        if out is None:
            out = np.zeros(max(1, total), dtype=np.uint32)
        ptr, cap = (out.data_ptr(), out.numel()) if hasattr(out, "data_ptr") else (out.ctypes.data, out.size)
        _check(L.tri_batch_docsets_mixed(self.h, ptr, cap, offs.ctypes.data, forms.ctypes.data))
        return out, offs, forms[: self.nq]

    def docset_form(self, q):
        """1 when the engine holds query q's docID set as a bitmap (RESULT_BITMAP), 0: ascending docIDs."""
        form, first, nw = C.c_int(), C.c_uint32(), C.c_size_t()
        _check(hip_lib().tri_batch_docset_bitmap(self.h, q, C.byref(form), None, 0, C.byref(first), C.byref(nw)))
        return form.value

    def docset_bitmap_words(self, q):
        """Words of query q's result bitmap (0: its docID set is held as ascending docIDs)."""
        form, first, nw = C.c_int(), C.c_uint32(), C.c_size_t()
        _check(hip_lib().tri_batch_docset_bitmap(self.h, q, C.byref(form), None, 0, C.byref(first), C.byref(nw)))
        return int(nw.value) if form.value else 0

    def docset_bitmap(self, q):
        """None when the engine holds query q's docID set as ascending docIDs; else (first_doc, words u32[]): bit j of words[i] = document
        first_doc + 32 i + j matches (DocumentsOnly queries expected to match one document in 32 or more; option result_bitmaps)."""
        L = hip_lib()
        form, first, nw = C.c_int(), C.c_uint32(), C.c_size_t()
        _check(L.tri_batch_docset_bitmap(self.h, q, C.byref(form), None, 0, C.byref(first), C.byref(nw)))
        if not form.value:
            return None
        words = np.zeros(nw.value, dtype=np.uint32)
        _check(L.tri_batch_docset_bitmap(self.h, q, C.byref(form), words.ctypes.data, nw.value, C.byref(first), C.byref(nw)))
        return first.value, words

    def matched_payloads(self, q):
        """FLAG_MATCHED_TERMS | FLAG_HIT_PAYLOADS batches: (lens u8[npos], payloads u64[npos]) parallel to matched_terms()'s positions."""
        L = hip_lib()
        n = C.c_size_t()
        _check(L.tri_batch_matched_payloads(self.h, q, None, None, 0, C.byref(n)))
        lens = np.zeros(max(1, n.value), dtype=np.uint8)
        pl = np.zeros(max(1, n.value), dtype=np.uint64)
        _check(L.tri_batch_matched_payloads(self.h, q, lens.ctypes.data, pl.ctypes.data, n.value, C.byref(n)))
        return lens[: n.value], pl[: n.value]

    def scores(self, q, n=None):
        if n is None:
            n = int(self.counts()[q])
        out = np.zeros(n, dtype=np.float64)
        got = C.c_size_t()
        _check(hip_lib().tri_batch_scores(self.h, q, out.ctypes.data, n, C.byref(got)))
        return out[: got.value]

    def docset_hashes(self):
        out = np.zeros(self.nq, dtype=np.uint64)
        _check(hip_lib().tri_batch_docset_hashes(self.h, out.ctypes.data))
        return out

    def device_results(self):
        """Raw device pointers of the result blocks the multi-GPU gather sends: {"counts": u64[nq]} and, for AccumulatedScore top-K
        batches, {"docs": u32[nq][k], "scores": f32[nq][k], "topk_counts": u32[nq]}.  Valid after the run completed on the engine stream."""
        L = hip_lib()
        p = C.c_void_p()
        _check(L.tri_batch_counts_device(self.h, C.byref(p)))
        out = {"counts": p.value}
        if (self.flags & FLAG_ACCUMULATED_SCORE) and self.topk:
            d, s, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
            _check(L.tri_batch_topk_device(self.h, C.byref(d), C.byref(s), C.byref(c)))
            out.update(docs=d.value, scores=s.value, topk_counts=c.value)
        return out

    def topk_results(self):
        d = np.zeros((self.nq, self.topk), dtype=np.uint32)
        s = np.zeros((self.nq, self.topk), dtype=np.float32)
        c = np.zeros(self.nq, dtype=np.uint32)
        _check(hip_lib().tri_batch_topk(self.h, d.ctypes.data, s.ctypes.data, c.ctypes.data))
        return d, s, c

    def close(self):
        if self.h:
            hip_lib().tri_batch_destroy(self.h)
            self.h = C.c_void_p()

class CollectionBatch:
    """The same queries over the sources of a collection (IndexSourcesCollection): one Batch per source, oldest first, each index
    carrying the documents the newer sources update as its masked set; counts add up, top-K lists merge on the device."""

    def __init__(self, batches, allow_unsupported=False):
        self.parts = list(batches)
        self.nq, self.topk = self.parts[0].nq, self.parts[0].topk
        arr = (C.c_void_p * len(self.parts))(*[b.h for b in self.parts])
        self.h = C.c_void_p()
        _check(hip_lib().tri_cbatch_create(arr, len(self.parts), C.byref(self.h)))
        if not allow_unsupported and self.query_status().any():
            self.close()
            raise TrinityError("queries of the collection batch have a shape the planner does not lower (allow_unsupported=True + query_status())")

    def query_status(self):
        out = np.zeros(self.nq, dtype=np.int32)
        _check(hip_lib().tri_cbatch_query_status(self.h, out.ctypes.data))
        return out

    def run(self):
        _check(hip_lib().tri_cbatch_run(self.h))

    def sync(self):
        _check(hip_lib().tri_cbatch_sync(self.h))

    def counts(self):
        out = np.zeros(self.nq, dtype=np.uint64)
        _check(hip_lib().tri_cbatch_match_counts(self.h, out.ctypes.data))
        return out

    def topk_results(self):
        d = np.zeros((self.nq, self.topk), dtype=np.uint32)
        s = np.zeros((self.nq, self.topk), dtype=np.float32)
        c = np.zeros(self.nq, dtype=np.uint32)
        _check(hip_lib().tri_cbatch_topk(self.h, d.ctypes.data, s.ctypes.data, c.ctypes.data))
        return d, s, c

    def docset(self, q, n):
        out = np.zeros(max(1, n), dtype=np.uint32)
        got = C.c_size_t()
        _check(hip_lib().tri_cbatch_docset(self.h, q, out.ctypes.data, n, C.byref(got)))
        return out[: got.value]

    def close(self):
        if self.h:
            hip_lib().tri_cbatch_destroy(self.h)
            self.h = C.c_void_p()
