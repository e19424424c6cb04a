"""FastPFor<4> payload words of the reference's lucene_codec (lucene_codec.cpp:57-64, 91-95) — restated from the library's published
algorithm because lemire/FastPFor is absent from the reference tree: PARITY UNPINNED against the genuine library.  Checked
here: two known answers derived by hand from the layout, the product's restatement (csrc/fastpfor128.hpp, through libtrinity_host.so) against
the oracle's (oracle/fastpfor128.c) word for word, and a segment written with these words against the same corpus in PFOR128: the
upload-time transcription yields the same directory."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import trinity_amd as T
from trinity_amd import hostplan as HP
from trinity_amd.engine import host_lib


def enc_product(v):
    L = host_lib()
    out = np.zeros(160, dtype=np.uint32)
    L.tri_host_fastpfor_encode.restype = C.c_uint32
    L.tri_host_fastpfor_encode.argtypes = [C.c_void_p, C.c_void_p]
    n = L.tri_host_fastpfor_encode(np.ascontiguousarray(v, dtype=np.uint32).ctypes.data, out.ctypes.data)
    return out[:n].copy()


def dec_product(w):
    L = host_lib()
    L.tri_host_fastpfor_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    v = np.zeros(128, dtype=np.uint32)
    ok = L.tri_host_fastpfor_decode(np.ascontiguousarray(w, dtype=np.uint32).ctypes.data, len(w), v.ctypes.data)
    return v if ok else None


def enc_oracle(v):
    f = O.lib().to_fastpfor_encode128
    f.restype = C.c_uint32
    f.argtypes = [C.c_void_p, C.c_void_p]
    out = np.zeros(160, dtype=np.uint32)
    n = f(np.ascontiguousarray(v, dtype=np.uint32).ctypes.data, out.ctypes.data)
    return out[:n].copy()


def dec_oracle(w):
    f = O.lib().to_fastpfor_decode128
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    v = np.zeros(128, dtype=np.uint32)
    return v if f(np.ascontiguousarray(w, dtype=np.uint32).ctypes.data, len(w), v.ctypes.data) else None


@pytest.fixture(scope="module", autouse=True)
def built():
    T.build.build_host()
    O.lib()


def test_known_answers_derived_from_the_layout():
    # 127 values of width 2 and one of width 3: b = 2 (cost 256 + 1 * (8 + 1) + 8 - 1 = 272 < 384), one exception of width maxb - b = 1 — its
    # high part is implied, nothing follows the (empty) bitmap
    v = np.full(128, 3, dtype=np.uint32)
    v[7] = 4
    want = [128, 9, 0xFFFF3FFF] + [0xFFFFFFFF] * 7 + [4, 0x07030102, 0]
    assert enc_product(v).tolist() == want == enc_oracle(v).tolist()
    # a 9-bit outlier among one-bit values: b = 1 (128 + 1 * (8 + 8) + 8 = 152 is the cheapest), the exception's high part 0x1f5 >> 1 = 0xfa at
    # width 8 in its own packed group of 32 (eight words), announced by bit 7 of the bitmap
    v = np.ones(128, dtype=np.uint32)
    v[0], v[10] = 0, 0x1F5
    want = [128, 5, 0xFFFFFFFE, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 4, 0x0A090101, 0x80, 1, 0xFA, 0, 0, 0, 0, 0, 0, 0]
    assert enc_product(v).tolist() == want == enc_oracle(v).tolist()
    assert dec_product(want).tolist() == v.tolist() == dec_oracle(want).tolist()


def test_product_and_oracle_restatements_agree():
    rng = np.random.default_rng(17)
    n = 0
    for trial in range(600):
        kind = trial % 6
        if kind == 0:
            v = rng.geometric(0.02, 128)  # document deltas of a mid-frequency term
        elif kind == 1:
            v = rng.integers(1, 4, 128)  # frequencies
        elif kind == 2:
            v = rng.integers(0, 1 << int(rng.integers(1, 33)), 128, dtype=np.uint64)
        elif kind == 3:
            v = rng.integers(0, 8, 128)
            v[rng.integers(0, 128, int(rng.integers(1, 40)))] = rng.integers(0, 1 << 32, dtype=np.uint64)  # heavy outliers: every exception width
        elif kind == 4:
            v = np.full(128, int(rng.integers(1, 1 << 20)))
            v[int(rng.integers(0, 128))] += 1  # almost constant (the all-equal form is the caller's: lucene_codec.cpp:31-39)
        else:
            v = rng.integers(0, 2, 128) << int(rng.integers(0, 32))
        v = np.asarray(v, dtype=np.uint64).astype(np.uint32)
        a, b = enc_product(v), enc_oracle(v)
        assert a.tolist() == b.tolist(), (trial, v.tolist())
        assert len(a) <= 255 and a[0] == 128  # the word count fits the ints() length byte
        assert dec_product(b).tolist() == v.tolist() == dec_oracle(a).tolist(), trial
        n += 1
        # damaged streams are refused, not misread: a cut, a wrong count, a position out of order
        assert dec_product(a[:-1]) is None and dec_oracle(a[:-1]) is None
        c = a.copy()
        c[0] = 127
        assert dec_product(c) is None and dec_oracle(c) is None
    assert n == 600



def test_a_value_count_near_2_32_is_refused_not_followed():
    """w[1] = 0xfffffffd passed `(n - 1) % 4 == 0` and `1 + n + 2 > L` wrapped to 0 in 32 bits: the decoder then
    read 16 GB past the group.  Both restatements must refuse such a group (TRI_ERR_FORMAT at upload), not crash."""
    for n in (0xFFFFFFFD, 0xFFFFFFF9, 0x7FFFFFFD, 125):
        w = np.array([128, n, 0, 0, 0, 0, 0, 0], dtype=np.uint32)
        assert dec_product(w) is None
        assert dec_oracle(w) is None
    # a byte count of the exception header that does not fit the group
    w = np.array([128, 1, 0xFFFFFFFE, 0, 0, 0, 0, 0], dtype=np.uint32)
    assert dec_product(w) is None and dec_oracle(w) is None

def _facts(h, nterms):
    L = host_lib()
    L.tri_host_index_facts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    info = np.zeros(6, dtype=np.uint64)
    per = np.zeros((nterms, 3), dtype=np.uint32)
    L.tri_host_index_facts(h.h, info.ctypes.data, per.ctypes.data)
    return info, per


def test_a_segment_with_fastpfor_words_transcodes_to_the_same_directory():
    D, V = 60_000, 3_000
    segs = {c: T.Segment(D, V, 10, 42, codec=c) for c in (2, 3)}
    assert segs[3].index.size != segs[2].index.size and not np.array_equal(segs[3].hits[:4096], segs[2].hits[:4096])
    assert np.array_equal(segs[2].terms[:, 0], segs[3].terms[:, 0])
    h2, h3 = HP.HostIndex.from_segment(segs[2]), HP.HostIndex.from_segment(segs[3])
    i2, p2 = _facts(h2, V)
    i3, p3 = _facts(h3, V)
    assert i2[4] == 0 and i2[5] == 0 and i3[4] > 1000 and i3[5] > 0  # the FastPFor-flavoured segment was transcribed group by group
    assert np.array_equal(p2, p3) and i2[0] == i3[0] and i2[1] == i3[1]  # documents, blocks, last documents: the same lists
    assert i3[2] != i2[2]  # ... while the SURVEY §8(d) byte counts are those of the bytes handed over
    # and the planner sees the same index
    parts, _ = HP_parts(D, V)
    for pt in parts:
        a = HP.HostPlan(h2, pt.programs, pt.flags, pt.topk, threads=2)
        b = HP.HostPlan(h3, pt.programs, pt.flags, pt.topk, threads=2)
        assert np.array_equal(a.tasks, b.tasks) and np.array_equal(a.sched, b.sched)


def HP_parts(D, V):
    from trinity_amd import workloads as W

    return W.build_parts("cfg3", D, V, 10, 42, 400)


# ------------------------------------------------------------------------------------------ the device encoder's units, on the CPU
def lucene_postings(rng, nterms):
    """Postings that reach every corner of the Lucene-shaped encoder: terms of 0 / 1 / 127 / 128 / 129 / 255 / 256 / 257 documents (blocks exactly full,
    one short, one over), hit counts that end exactly on a 128-hit block, deltas of every varint length, a document of 700 hits."""
    docs, freqs, pos, tf = [], [], [], [0]
    sizes = [0, 1, 127, 128, 129, 255, 256, 257, 1000, 5, 384, 640]
    for t in range(nterms):
        n = sizes[t % len(sizes)] if t < 3 * len(sizes) else int(rng.integers(0, 700))
        d = np.cumsum(rng.integers(1, [1, 3, 200, 20000, 3_000_000][t % 5] + 1, size=n, dtype=np.int64))
        d = d[d < 2**32 - 1]
        f = rng.integers(0, [2, 4, 9, 40][t % 4], size=d.size)
        if d.size and t % 7 == 0:
            f[int(rng.integers(0, d.size))] = 700
        if d.size and t % 9 == 0:  # (the term's hits end exactly on a block of 128)
            short = (-int(f.sum())) % 128
            f[-1] += short
        for k in f.tolist():
            pos += np.sort(rng.integers(1, [12, 200, 65536][t % 3], size=k)).tolist()
        docs += d.tolist()
        freqs += f.tolist()
        tf.append(len(docs))
    return np.array(docs, np.uint32), np.array(freqs, np.uint32), np.array(pos, np.uint16), np.array(tf, np.uint64)


def test_the_device_encoders_group_writer_equals_the_host_encoders():
    """csrc/pfor128_group.hpp (plan + emit through a getter: what a lane of k_lencode.hpp runs) against lucene_encoder.hpp::ints_encode, byte for byte:
    all-equal groups, every width, exceptions of every count and size, the five-byte varint."""
    T.build.build_host()
    rng = np.random.default_rng(1)
    for trial in range(6000):
        kind = trial % 7
        if kind == 0:
            v = np.full(128, rng.integers(0, 2**32), np.uint32)
        elif kind == 1:
            v = rng.integers(0, 2 ** int(rng.integers(1, 33)), 128, dtype=np.uint64).astype(np.uint32)
        elif kind == 2:
            v = rng.integers(0, 2 ** int(rng.integers(1, 12)), 128, dtype=np.uint64).astype(np.uint32)
            k = int(rng.integers(1, 40))
            v[rng.choice(128, k, replace=False)] = rng.integers(0, 2**32, k, dtype=np.uint64).astype(np.uint32)
        elif kind == 3:
            v = np.zeros(128, np.uint32)
            v[int(rng.integers(0, 128))] = int(rng.integers(1, 2**32))
        elif kind == 4:
            v = rng.integers(0, 2, 128).astype(np.uint32)
        elif kind == 5:
            v = rng.integers(1, 300, 128).astype(np.uint32)
        else:
            v = (rng.integers(0, 2**32, 128, dtype=np.uint64) >> rng.integers(0, 32, 128).astype(np.uint64)).astype(np.uint32)
        a, b = HP.pfor128_group_pair(v)
        assert a is not None and np.array_equal(a, b), (trial, kind)


def test_the_device_encoders_units_equal_the_sequential_encoder():
    """csrc/lucene_enc_units.hpp — per-block sizes, prefix sums, per-block and per-term writes, the skiplist entries from the postings' indices — run in plain
    loops, against the sequential state machine of lucene_encoder.hpp: `index`, `hits.data` and the term table, byte for byte.  (The kernels of k_lencode.hpp
    are these units, one per lane: tests/test_gpu_parity.py::test_lucene_encoder_on_the_device.)"""
    T.build.build_host()
    rng = np.random.default_rng(3)
    for nterms in (1, 12, 36, 300):
        d, f, p, tf = lucene_postings(rng, nterms)
        want = HP.lucene_encode(d, f, p, tf)
        got = HP.lucene_encode(d, f, p, tf, units=True)
        for x, y, name in zip(got, want, ("index", "hits.data", "terms")):
            assert x.shape == y.shape and np.array_equal(x, y), (nterms, name)
    assert want[0].size > 100_000 and want[1].size > 300_000
