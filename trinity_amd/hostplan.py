"""The host planner without a device (csrc/host/plan_host.cpp in libtrinity_host.so): plans a batch exactly as tri_batch_create does
and hands back the plan's host block — for the CPU tests of the planner (tests/test_planner.py) and tools/plan_probe.py.  Test and
probe infrastructure: the product path is tri_batch_create in libtrinity_hip.so, which includes the same planner.hpp."""
import ctypes as C

import numpy as np

from .engine import TrinityError, flatten, host_lib  # noqa: F401

_SUMMARY = ["block_bytes", "n_plan", "n_qterms", "n_tasks", "n_fused_maps", "n_qplane", "n_plane_terms", "n_sterms", "n_sweights", "n_phrases", "n_pterms", "n_ptasks",
            "off_plan", "off_qterms", "off_tasks", "off_sched", "off_fused", "off_qplane", "off_plane_terms", "off_sterms", "off_sweights", "off_phrases", "off_pterms", "off_ptasks",
            "n_dense", "n_cand", "n_fused", "n_fused16", "n_fusedgen", "n_planes", "n_planes8", "plw", "sparse_cap", "out_capacity", "term_bytes", "term_bytes_dense",
            "dense_queries", "cand_queries", "fused_queries", "planes_queries", "unsupported_queries", "rich_R", "sizeof_query", "sizeof_task", "sizeof_fused", "sizeof_phrase",
            "cand_needed_term_bytes", "plane_decoded_bytes", "n_pset", "pset_queries", "n_probe", "probe_queries", "n_units", "off_units", "off_unit_sched", "sizeof_unit",
            "n_tree", "tree_queries", "n_tree_words", "off_tree", "n_tree_terms", "off_tree_terms", "n_tree_hidden", "off_tree_hidden", "off_cand_q"]  # fmt: skip

DEV_QUERY = np.dtype([("nterms", "<u4"), ("term_base", "<u4"), ("out_off", "<u8"), ("out_cap", "<u4"), ("qid", "<u4"), ("first_task", "<u4"), ("ntasks", "<u4"),
                      ("score_base", "<u4"), ("nscore", "<u4"), ("phrase_base", "<u4"), ("nphrases", "<u4"), ("fused_idx", "<u4"), ("form", "<u4")])  # fmt: skip
DEV_TASK = np.dtype([("slot", "<u4"), ("begin", "<u4"), ("end", "<u4"), ("kind", "<u4"), ("out_off", "<u8")])
TASK_CAND, TASK_DENSE, TASK_FUSED, TASK_FUSED16, TASK_FUSED_GEN, TASK_PLANES, TASK_PLANES8, TASK_PSET, TASK_PROBE, TASK_TREE = range(10)
DEV_TREE_NODE = np.dtype([("op", "u1"), ("parent", "u1"), ("ord", "u1"), ("thr", "u1"), ("arg", "<u4"), ("row", "<u4"), ("score", "<u4"), ("rmask", "<u4"),
                          ("kid0", "u1"), ("kid1", "u1"), ("pad", "u1", 2), ("kids", "<u8")])  # fmt: skip
TREE_HDR_WORDS = 8
DEV_UNIT = np.dtype([("out_off", "<u8"), ("begin", "<u4"), ("end", "<u4"), ("tix", "<u4"), ("nterms", "<u4"), ("term_base", "<u4"), ("first", "<u4"), ("tt", "<u4", 4), ("row", "<u4", 4)])
SCHED_ORDER = [TASK_DENSE, TASK_PSET, TASK_PROBE, TASK_CAND, TASK_FUSED, TASK_FUSED16, TASK_FUSED_GEN, TASK_PLANES, TASK_PLANES8, TASK_TREE]


def _lib():
    L = host_lib()
    if not hasattr(L, "_plan_ready"):
        vp = C.c_void_p
        L.tri_host_index_build.restype = vp
        L.tri_host_index_build.argtypes = [vp, C.c_uint64, vp, C.c_uint64, C.c_int, vp, C.c_uint64, C.c_uint32, C.c_char_p, C.c_uint64]
        L.tri_host_index_free.argtypes = [vp]
        L.tri_host_plan.restype = vp
        L.tri_host_plan.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_uint, vp, vp, C.c_uint, C.c_uint32, C.c_char_p, C.c_uint64]
        L.tri_host_plan_free.argtypes = [vp]
        L.tri_host_plan_summary.argtypes = [vp, vp, vp]
        L.tri_host_plan_block.restype = C.POINTER(C.c_uint8)
        L.tri_host_plan_block.argtypes = [vp]
        L.tri_host_plan_query_maps.argtypes = [vp, vp, vp]
        L.tri_host_pfor128_group.argtypes = [vp, vp, C.POINTER(C.c_uint32), vp, C.POINTER(C.c_uint32)]
        L.tri_host_pfor128_group.restype = None
        for f in (L.tri_host_lucene_encode, L.tri_host_lucene_encode_units):
            f.argtypes = [vp, vp, vp, vp, C.c_uint64, vp, C.c_uint64, C.POINTER(C.c_uint64), vp, C.c_uint64, C.POINTER(C.c_uint64), vp]
        L._plan_ready = True
    return L


class HostIndex:
    def __init__(self, index_bytes, terms, docs_cnt, codec=1, hits=None):
        b = np.ascontiguousarray(index_bytes, dtype=np.uint8)
        t = np.ascontiguousarray(terms, dtype=np.uint32).reshape(-1, 3)
        h = np.ascontiguousarray(hits, dtype=np.uint8) if hits is not None and len(hits) else None
        err = C.create_string_buffer(600)
        self.h = _lib().tri_host_index_build(b.ctypes.data, b.size, h.ctypes.data if h is not None else None, h.size if h is not None else 0, codec, t.ctypes.data, t.shape[0],
                                             docs_cnt, err, 600)  # fmt: skip
        if not self.h:
            raise TrinityError(err.value.decode())

    @classmethod
    def from_segment(cls, seg):
        return cls(seg.index, seg.terms, seg.docs_cnt, codec=seg.codec, hits=seg.hits)

    def close(self):
        if self.h:
            _lib().tri_host_index_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class HostPlan:
    """One planned batch: .s (summary dict), .ms (the four phase times), and the plan's arrays as numpy views of (a copy of) its block."""

    def __init__(self, hindex, programs, flags, topk=0, similarity=0, threads=1, options=None, cus=256, flat=None):
        self.nq = len(programs) if flat is None else flat[1].shape[0]
        prog, q = flatten(programs) if flat is None else flat
        options = options or {}
        names = (C.c_char_p * max(1, len(options)))(*[k.encode() for k in options])
        values = np.array(list(options.values()) or [0], dtype=np.uint64)
        err = C.create_string_buffer(600)
        self.h = _lib().tri_host_plan(hindex.h, prog.ctypes.data, prog.size, q.ctypes.data, self.nq, flags, topk, similarity, threads, names, values.ctypes.data, len(options), cus,
                                      err, 600)  # fmt: skip
        if not self.h:
            raise TrinityError(err.value.decode())
        out = np.zeros(len(_SUMMARY), dtype=np.uint64)
        ms = np.zeros(4, dtype=np.float64)
        _lib().tri_host_plan_summary(self.h, out.ctypes.data, ms.ctypes.data)
        self.s = {k: int(v) for k, v in zip(_SUMMARY, out)}
        self.ms = ms
        assert self.s["sizeof_query"] == DEV_QUERY.itemsize and self.s["sizeof_task"] == DEV_TASK.itemsize
        p = _lib().tri_host_plan_block(self.h)
        # (a copy of its own: the arrays handed out below are views of it and stay readable after close() — a failing test's report formats them then)
        self.block = np.ctypeslib.as_array(p, shape=(max(1, self.s["block_bytes"]),)).copy()
        self.slot_of_query = np.zeros(max(1, self.nq), dtype=np.uint32)
        self.qstatus = np.zeros(max(1, self.nq), dtype=np.int32)
        _lib().tri_host_plan_query_maps(self.h, self.slot_of_query.ctypes.data, self.qstatus.ctypes.data)

    def _view(self, off, n, dtype):
        dt = np.dtype(dtype)
        return self.block[self.s[off] : self.s[off] + n * dt.itemsize].view(dt)

    @property
    def plan(self):
        return self._view("off_plan", self.s["n_plan"], DEV_QUERY)

    @property
    def tasks(self):
        return self._view("off_tasks", self.s["n_tasks"], DEV_TASK)

    @property
    def sched(self):
        return self._view("off_sched", self.s["n_tasks"], "<u4")

    @property
    def units(self):
        return self._view("off_units", self.s["n_units"], DEV_UNIT)

    @property
    def unit_sched(self):
        return self._view("off_unit_sched", self.s["n_pset"] + self.s["n_probe"], "<u4")

    @property
    def cand_q(self):
        """k_and's queues, one per XCD: queue x is the TASK_CAND section of sched at [cand_q[x], cand_q[x + 1])."""
        return self._view("off_cand_q", 9, "<u4")

    @property
    def qterms(self):
        return self._view("off_qterms", self.s["n_qterms"], "<u4")

    @property
    def qplane(self):
        return self._view("off_qplane", self.s["n_qplane"], "<u4")

    @property
    def plane_terms(self):
        return self._view("off_plane_terms", self.s["n_plane_terms"], "<u4")

    @property
    def tree_terms(self):
        return self._view("off_tree_terms", self.s["n_tree_terms"], "<u4")

    @property
    def tree_hidden(self):
        return self._view("off_tree_hidden", self.s["n_tree_hidden"], "<u4")

    def tree_nodes(self, slot):
        """The DevTreeNode records of the TASK_TREE query in plan slot `slot`."""
        words = self._view("off_tree", self.s["n_tree_words"], "<u4")
        at = int(self.plan[slot]["fused_idx"])
        n = int(words[at])
        return words[at + TREE_HDR_WORDS : at + TREE_HDR_WORDS + 8 * n].view(DEV_TREE_NODE)

    def close(self):
        if self.h:
            self.block = None
            _lib().tri_host_plan_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def pfor128_group_pair(v):
    """One ints() group of 128 values two ways: (the device encoder's plan / emit pair, the host encoder) -> (bytes, bytes)."""
    v = np.ascontiguousarray(v, dtype=np.uint32)
    assert v.size == 128
    a, b = np.zeros(700, dtype=np.uint8), np.zeros(700, dtype=np.uint8)
    al, bl = C.c_uint32(), C.c_uint32()
    _lib().tri_host_pfor128_group(v.ctypes.data, a.ctypes.data, C.byref(al), b.ctypes.data, C.byref(bl))
    return a[: al.value] if al.value != 0xFFFFFFFF else None, b[: bl.value]


def lucene_encode(docs, freqs, positions, term_first, units=False):
    """The Lucene-shaped encoder on the host: the sequential encoder (csrc/host/lucene_encoder.hpp), or — units=True — the device encoder's units
    (csrc/lucene_enc_units.hpp) run in plain loops.  -> (index bytes, hits.data bytes, term table u32[n, 3])."""
    d = np.ascontiguousarray(docs, dtype=np.uint32)
    f = np.ascontiguousarray(freqs, dtype=np.uint32)
    p = np.ascontiguousarray(positions, dtype=np.uint16)
    tf = np.ascontiguousarray(term_first, dtype=np.uint64)
    n = tf.size - 1
    icap, hcap = 128 + 16 * n + 12 * d.size, 128 + 6 * p.size + 8 * n
    io, ho, t3 = np.zeros(icap, dtype=np.uint8), np.zeros(hcap, dtype=np.uint8), np.zeros((max(n, 1), 3), dtype=np.uint32)
    il, hl = C.c_uint64(), C.c_uint64()
    fn = _lib().tri_host_lucene_encode_units if units else _lib().tri_host_lucene_encode
    if fn(d.ctypes.data, f.ctypes.data, p.ctypes.data, tf.ctypes.data, n, io.ctypes.data, icap, C.byref(il), ho.ctypes.data, hcap, C.byref(hl), t3.ctypes.data):
        raise TrinityError("lucene_encode: buffer too small")
    return io[: il.value], ho[: hl.value], t3[:n]


def cpu_budget():
    """The CPUs this process may use at once as the planner sees them (csrc/host_pool.hpp host_cpu_budget): the affinity mask, capped by the cgroup's CPU quota,
    divided by LOCAL_WORLD_SIZE under a one-process-per-GPU launcher."""
    L = host_lib()
    L.tri_host_cpu_budget.restype = C.c_uint32
    return int(L.tri_host_cpu_budget())


def pool_cpus(threads=16):
    """The CPUs the planner's host pool of `threads` threads pins its workers to in this process (csrc/host_pool.hpp: a rank's own slice of the
    affinity mask when LOCAL_RANK / LOCAL_WORLD_SIZE are set, else the CPUs next to the calling thread's)."""
    L = host_lib()
    L.tri_host_pool_cpus.restype = C.c_uint32
    L.tri_host_pool_cpus.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32]
    out = np.zeros(256, dtype=np.int32)
    n = L.tri_host_pool_cpus(threads, out.ctypes.data, out.size)
    return out[:n].tolist()
