"""CPU tests: the C-ABI library builds, loads and exports every symbol include/trinity_hip.h declares; the
host-side segment builder reproduces the oracle's (== the reference's) bytes.  No GPU compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def T():
    import trinity_amd

    trinity_amd.build_all()
    return trinity_amd


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "trinity_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tri_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(T):
    from trinity_amd.engine import ABI_SYMBOLS, hip_lib

    decl = declared_symbols()
    assert decl == sorted(ABI_SYMBOLS), set(decl) ^ set(ABI_SYMBOLS)
    L = hip_lib()
    for s in decl:
        assert getattr(L, s) is not None
    assert L.tri_abi_version() == 9


def test_missing_library_fails_loudly(T, monkeypatch):
    import trinity_amd.engine as E

    monkeypatch.setattr(E, "_hip", None)
    monkeypatch.setattr(E, "LIB_HIP", "/nonexistent/libtrinity_hip.so")
    with pytest.raises(E.TrinityError):
        E.hip_lib()


def test_product_path_never_touches_oracle():
    """Nothing under trinity_amd/ or include/ may reference oracle/ (the checker) or a CPU fallback."""
    bad = []
    for base in ("trinity_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle_lib|liboracle|trinity_oracle|#include\s+\"[^\"]*oracle/", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


@pytest.mark.parametrize("cfg", [(2000, 200, 10, 42), (20000, 500, 12, 7), (30000, 5000, 10, 3)])
def test_segment_builder_matches_oracle_bytes(T, cfg):
    D, V, S, seed = cfg
    seg = T.Segment(D, V, S, seed)
    ix = O.Index.generate(D, V, S, seed)
    assert np.array_equal(seg.index, ix.bytes())
    assert np.array_equal(seg.terms, ix.terms())
    assert (seg.sum_terms_docs, seg.total_terms, seg.docs_cnt) == (int(ix.c.sumTermsDocs), int(ix.c.totalTerms), int(ix.c.docsCnt))


def test_query_generator_matches_oracle(T):
    assert np.array_equal(T.gen_queries(10000, 1337, 200, 2), O.gen_queries(10000, 1337, 200, 2))
    assert np.array_equal(T.gen_queries(777, 5, 50, 5), O.gen_queries(777, 5, 50, 5))
    q = T.gen_queries(50, 9, 100, 5)
    assert all(len(set(r)) == 5 for r in q.tolist())


def test_no_gpu_means_error_not_fallback(T):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(T.TrinityError):
        T.Device(0)


@pytest.mark.parametrize("cfg", [(2000, 200, 10, 42), (20000, 500, 12, 7), (30000, 5000, 10, 3)])
def test_lucene_segment_builder_matches_oracle_bytes(T, cfg):
    """Two independent writers of the Lucene-shaped container + PFOR128 payload (include/pfor128.md) agree byte for byte."""
    from trinity_amd.engine import CODEC_LUCENE

    D, V, S, seed = cfg
    seg = T.Segment(D, V, S, seed, codec=CODEC_LUCENE)
    ix = O.Index.generate(D, V, S, seed, codec="lucene")
    assert np.array_equal(seg.index, ix.bytes()) and np.array_equal(seg.hits, ix.hits()) and np.array_equal(seg.terms, ix.terms())


def test_error_contract_without_a_gpu():
    """C-ABI error behaviour (include/trinity_hip.h: int status + tri_last_error, nothing thrown across the boundary).  Runs on
    the CPU-only build box as well: without a device tri_dev_open must FAIL loudly (there is no fallback), and argument
    errors are reported before any device work."""
    import ctypes as C

    import trinity_amd.engine as E

    L = E.hip_lib()
    dev = C.c_void_p()
    rc = L.tri_dev_open(10_000, C.byref(dev))  # no such device anywhere
    assert rc < 0 and not dev.value
    assert L.tri_last_error()  # a message is always left behind
    out = C.c_void_p()
    assert L.tri_batch_create(None, None, 0, None, 0, None, 1, 0, 0, C.byref(out)) == -1 and not out.value  # TRI_ERR_INVALID
    assert L.tri_index_upload(None, None, 0, None, 0, 1, None, 0, 0, C.byref(out)) == -1
    assert L.tri_batch_run(None) < 0 and L.tri_batch_sync(None) < 0
    n = C.c_size_t()
    assert L.tri_batch_docset(None, 0, None, 0, C.byref(n)) < 0
    assert L.tri_index_set_masked(None, None, 0) < 0
    L.tri_batch_destroy(None)  # destroying nothing is a no-op
    L.tri_index_destroy(None)
    L.tri_dev_close(None)
    import pytest

    import torch

    if not torch.cuda.is_available():
        with pytest.raises(E.TrinityError):
            E.Device(0)  # the Python binding turns the status into an exception; it never falls back to a CPU path
