#!/usr/bin/env python3
"""tri_batch_create timed in a process that has an RCCL communicator (torch.distributed, backend nccl) against one that has not:
   python tools/probe_create_dist.py            (plain)
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 tools/probe_create_dist.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import trinity_amd as T
from trinity_amd import workloads as W

def t_create(ix, progs, flags, topk, n=3):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); b = T.Batch(ix, progs, flags, topk=topk); dt = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter(); b.close(); dc = (time.perf_counter() - t0) * 1e3
        best = min(best, dt)
    return best, dc

# cfg5's DocumentsOnly part at full size: its OR-5 unions bound-allocate some 15 GB of output (the allocation is the point here)
parts, desc = W.build_parts("cfg5", 10_000_000, 1_000_000, 10, 42, 12500)
progs, flags, topk, codec = parts[0].programs, parts[0].flags, parts[0].topk, parts[0].codec
seg = T.Segment(10_000_000, 1_000_000, 10, 42, codec=codec)
dev = T.Device(0)
ix = T.Index.from_segment(dev, seg)
print("env OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"), "affinity", len(os.sched_getaffinity(0)), flush=True)
print("before any torch.cuda / dist: create %.2f ms close %.2f ms" % t_create(ix, progs, flags, topk), flush=True)
torch.cuda.set_device(0); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
print("after torch.cuda init:        create %.2f ms close %.2f ms" % t_create(ix, progs, flags, topk), flush=True)
if "RANK" in os.environ:
    import torch.distributed as dist
    dist.init_process_group("nccl")
    print("after init_process_group:     create %.2f ms close %.2f ms" % t_create(ix, progs, flags, topk), flush=True)
    t = torch.ones(1, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    print("after the first collective:   create %.2f ms close %.2f ms" % t_create(ix, progs, flags, topk), flush=True)
    dist.destroy_process_group()
