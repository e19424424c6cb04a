#!/usr/bin/env python3
"""Which neighbour makes one tri_batch_create in a hundred take 10 - 50 ms?  (Round 6's answer: the container's CPU quota — cgroup cpu.max 16 CPUs on a box that shows
256: two pools of 15 polling workers got the whole process throttled.  The planner now sizes its pools to host_cpu_budget(), host_pool.hpp.)  cfg2, 16384 queries, N creates back to back on a compiler thread while the main thread
(i) sleeps, (ii) runs + awaits a resident batch over and over, (iii) also reads its match counts back, (iv) as (iii) with a second compiler thread.
Prints per variant the creates' median / p99 / max and how many took over 3 ms.    python tools/probe_create_jitter.py [N]"""
import os, sys, time, threading, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import trinity_amd as T
from trinity_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
parts, desc = W.build_parts("cfg2", 10_000_000, 1_000_000, 10, 42, 16384)
pt = parts[0]
seg = T.Segment(10_000_000, 1_000_000, 10, 42, codec=pt.codec)
dev = T.Device(0)
ix = T.Index.from_segment(dev, seg)
flat = T.engine.flatten(pt.programs)
mk = lambda: T.Batch(ix, None, pt.flags, topk=pt.topk, flat=flat)
print('cpu.max:', open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'n/a', flush=True)
res = mk(); res.run(); res.sync(); res.counts()
for _ in range(8):
    b = mk(); b.close()

def creates(k, out):
    for _ in range(k):
        t0 = time.perf_counter(); b = mk(); out.append((time.perf_counter() - t0) * 1e3); b.close()

def throttled():
    try:
        st = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(st.get("nr_throttled", 0)), int(st.get("throttled_usec", 0))
    except OSError:
        return 0, 0


def variant(tag, main_mode, ncomp):
    th0 = throttled()
    outs = [[] for _ in range(ncomp)]
    ths = [threading.Thread(target=creates, args=(n // ncomp, outs[i])) for i in range(ncomp)]
    for th in ths: th.start()
    steps = 0
    while any(th.is_alive() for th in ths):
        if main_mode == 0:
            time.sleep(0.01)
        else:
            res.run(); res.sync(); steps += 1
            if main_mode == 2:
                res.counts()
    for th in ths: th.join()
    x = np.array(sum(outs, [])[5:])
    th1 = throttled()
    print(f"{tag:62s} creates {len(x)}: median {np.median(x):.3f} p99 {np.percentile(x, 99):.3f} max {x.max():.3f} ms; over 3 ms: {(x > 3).sum()}; main steps {steps}; "
          f"cgroup periods throttled {th1[0] - th0[0]} ({(th1[1] - th0[1]) / 1e3:.0f} ms of thread time)", flush=True)

variant("(i) one compiler, main sleeps", 0, 1)
variant("(ii) one compiler, main runs + syncs a resident batch", 1, 1)
variant("(iii) one compiler, main runs + syncs + reads counts back", 2, 1)
variant("(iv) two compilers, main runs + syncs + reads counts back", 2, 2)
variant("(v) two compilers, main sleeps", 0, 2)
