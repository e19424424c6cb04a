#!/usr/bin/env python3
"""PMC calibration workload for rocprofv3 FETCH_SIZE on the engine's own access pattern: the single query
`t0 AND t1` (TASK_DENSE) streams both chunks and their directory rows exactly once per launch with per-lane
8-byte loads, and writes |t0 ∩ t1| docIDs.  Known bytes are printed; run under
`rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv` (MI355X_MICROARCH.md §HBM: widths other than
16 B/lane are uncalibrated — this calibrates ours)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import trinity_amd as T
seg = T.Segment(10_000_000, 1_000_000, 10, 42)
dev = T.Device(0)
ix = T.Index.from_segment(dev, seg)
res = []
for q in ([0, 1], [2, 3]):
    b = T.Batch.conjunctions(ix, np.array([q], dtype=np.uint32))
    for _ in range(3):
        b.run(); b.sync()
    inf = b.info()
    chunk = sum(int(seg.terms[t, 2]) for t in q)
    blocks = sum((int(seg.terms[t, 0]) + 31) // 32 for t in q)
    res.append({"query": q, "chunk_bytes": chunk, "directory_bytes": blocks * 8, "read_bytes_expected": chunk + blocks * 8, "written_bytes": int(inf["matches"]) * 4, "kernel_ms": inf["last_run_ms"]})
    b.close()
print(json.dumps(res))
