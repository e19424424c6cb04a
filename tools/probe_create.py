import os, sys, time
sys.path.insert(0, os.getcwd())
import trinity_amd as T
from trinity_amd import workloads as W
for name in ("cfg2", "cfg3"):
    progs, flags, topk, codec, desc = W.build(name, 10_000_000, 1_000_000, 10, 42, 16384 if name == "cfg2" else 8192)
    seg = T.Segment(10_000_000, 1_000_000, 10, 42, codec=codec)
    dev = T.Device(0)
    ix = T.Index.from_segment(dev, seg)
    for acc in (0, 1):
        dev.set_option("account_needed_bytes", acc)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); b = T.Batch(ix, progs, flags, topk=topk); dt = (time.perf_counter() - t0) * 1e3; b.close()
            best = min(best, dt)
        print(name, "account_needed_bytes", acc, "create %.2f ms" % best, flush=True)
    ix.close(); dev.close()
