"""Debug driver (not a test): runs a few AND queries against a chosen engine build with the in-kernel trace
watchdog.  usage: TRINITY_HIP_LIB=trinity_amd/libtrinity_hip_dbgA.so python tools/debug_and.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import trinity_amd.build as B
if os.environ.get("TRINITY_HIP_LIB"):
    import trinity_amd.engine as E
    E.LIB_HIP = os.path.abspath(os.environ["TRINITY_HIP_LIB"])
import trinity_amd as T
import oracle_lib as O

D, V = 20000, 2000
seg = T.Segment(D, V, 10, 42)
ora = O.Index.wrap(seg.index, seg.terms, seg.docs_cnt, seg.sum_terms_docs, seg.sum_term_hits)
dev = T.Device(0)
ix = T.Index.from_segment(dev, seg)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
sets = {"one": [[0, 1]], "rare": [[1500, 0]], "few": [[0, 1], [3, 700], [5, 9], [1000, 1500]], "all": T.gen_queries(V, 1337, 64, 2).tolist()}
qs = np.array(sets[which], dtype=np.uint32)
t0 = time.time()
b = T.Batch.conjunctions(ix, qs)
b.run(); b.sync()
counts = b.counts()
bad = 0
for i, (a, c) in enumerate(qs.tolist()):
    want, _ = ora.exec(np.array([T.tok(0, a), T.tok(0, c), T.tok(1, 2)], dtype=np.uint32), 1)
    got = b.docset(i, int(counts[i]))
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH", i, a, c, len(got), len(want), flush=True)
print(f"{which}: {len(qs)} queries, bad={bad}, {time.time()-t0:.2f}s, kernel {b.info()['last_run_ms']:.3f} ms", flush=True)
if os.environ.get("DBG_CLOSE"):
    print("hashes...", flush=True); h = b.docset_hashes(); print("hashes ok", flush=True)
    print("closing batch", flush=True); b.close(); print("batch closed", flush=True)
    ix.close(); print("index closed", flush=True)
    dev.close(); print("dev closed", flush=True)
