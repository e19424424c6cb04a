#!/usr/bin/env python3
"""bench.py — the hot path's headline benchmark on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Default workload (BASELINE.json configs[1]): batched 2-term AND queries (Zipf-sampled terms, seed 1337,
distinct within a query) over the 10M-document / 1M-term synthetic Zipf(1.0) segment (corpus seed 42, google_codec),
DocumentsOnly mode — every query's full ascending docID set is materialised in HBM.  --workload cfg3 / cfg4 / cfg5 / cfg1 are
the other query sets of SURVEY §8(d) (cfg5 = the mixed batch: a DocumentsOnly batch on the google_codec segment plus a BM25
top-100 batch on the lucene_codec segment of the same corpus, both per step).

One "step" = what the reference's exec_query does for every query of a batch of --queries queries per GPU, end to end
(exec.cpp:530-876, 1511-1516: plan the query, run it, deliver the results): tri_batch_create (host planning on the device handle's
host threads + one H2D copy of the plan) -> tri_batch_run -> tri_batch_sync -> read-back of the match counts (and, scored, the
top-K blocks) to the host [-> the result all_gather over RCCL at N > 1], PIPELINED: the next step's batch is compiled and uploaded
while the current one runs — the timed K steps hold exactly K creates, K runs, K read-backs.  The index is resident in HBM; the
docID sets themselves stay in HBM (DESIGN.md §5 has the PCIe-inclusive figure).  `kernels_only` is the same batch re-run without
planning or read-back (HIP-event time of the runs inside the same timed steps).

Multi-GPU: one process per GPU, the index replicated, the query stream sharded (queries are independent, exec.h:57-62):
--scaling weak gives every rank a shard of --queries queries, --scaling strong splits --queries over the ranks; with --gpus N > 1 the
default workload is cfg5, the mixed 100K-query batch BASELINE.json's scaling criterion is quoted on, STRONG scaling: the 100 000 queries
split over the N ranks (N = 1: weak, the cfg2 headline).  At the end of every step the ranks all_gather their result blocks over RCCL straight from the engine's device buffers:
per-query match counts and, for scored batches, the [Q/G][K] top-K docID/score blocks (trinity_amd/dist.py ResultGather — the
only exchange the path has).  Every line carries `scaling_point` — the mixed 100 K-query batch, strong scaling: value, per_gpu_value,
gather_ms, hbm_bytes_in_use and `n1`, the SAME workload on one GPU measured in the same run (N = 1: this GPU runs the whole 100 K batch after
the cfg2 headline; N > 1: rank 0 alone on the whole batch, the other ranks parked at a barrier) — so a curve can be read from the lines alone.

Rank 0 prints ONE JSON line: metric/value = queries/s over all GPUs.  `roofline`: bound "hbm"; `frac` is the BATCH-LEVEL bound — every
distinct list the step's queries name read once + every output written once (tri_batch_info.bound_bytes) / the step's kernel time /
peak — for the whole step and, under `kernels`, per kernel (HIP events on the engine's stream around each launch, inside the timed
steps), next to `physical_frac` (committed rocprofv3 PMC traffic of the same workload / kernel time / peak) and SURVEY §8(d)'s per-query
algorithmic bytes (`per_query_algorithmic`: a figure no kernel that shares decodes or skips reads, so it carries no fraction).
`cpu_baseline` = a bounded sample of the same batch on the host's cores: the CPU oracle (a restatement of the reference's exec path, planning
included; kind "port"), which doubles as a per-query full-size parity check, and — for the Google-codec workloads, where oracle/_ref/ref_driver
is there — the GENUINE reference's exec_query over the same queries (kind "reference": that figure leads, the port's stands under `port`).
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SCALING_TOTAL_QUERIES = 100_000  # the mixed batch north_star's scaling target is worded on (BASELINE.json configs[4])


def hbm_in_use(mem):
    """tri_dev_memory -> what the bench line states: the engine's pooled buffers in use (batches' arenas, output regions, plane caches), the idle ones it keeps,
    and everything resident on the device (indexes, RCCL, the runtime: total - free)."""
    return {"engine_pool_in_use": mem["pool_in_use_bytes"], "engine_pool_idle": mem["pool_idle_bytes"], "device_in_use": mem["device_total_bytes"] - mem["device_free_bytes"],
            "device_total": mem["device_total_bytes"]}
KERNELS = ("k_and_dense", "k_psets", "k_probe", "k_and", "k_fused", "k_planes", "k_phrase")  # the kernels with per-launch HIP-event brackets and their own byte counts
KMS = {"k_and_dense": "dense_ms", "k_psets": "pset_ms", "k_probe": "probe_ms", "k_and": "cand_ms", "k_fused": "fused_ms", "k_planes": "planes_ms", "k_phrase": "phrase_ms"}
KALG = {"k_and_dense": "dense_algorithmic_bytes", "k_psets": "pset_algorithmic_bytes", "k_probe": "probe_algorithmic_bytes", "k_and": "cand_algorithmic_bytes", "k_fused": "fused_algorithmic_bytes", "k_planes": "planes_algorithmic_bytes",
        "k_phrase": "phrase_algorithmic_bytes"}  # fmt: skip
KBOUND = {"k_and_dense": "dense_bound_bytes", "k_psets": "pset_bound_bytes", "k_probe": "probe_bound_bytes", "k_and": "cand_bound_bytes", "k_fused": "fused_bound_bytes", "k_planes": "planes_bound_bytes", "k_phrase": "phrase_bound_bytes"}
KQ = {"k_and_dense": "dense_queries", "k_psets": "pset_queries", "k_probe": "probe_queries", "k_and": "cand_queries", "k_fused": "fused_queries", "k_planes": "planes_queries", "k_phrase": "phrase_queries"}
MS_KEYS = ("last_run_ms", "dense_ms", "pset_ms", "probe_ms", "cand_ms", "fused_ms", "phrase_ms", "rest_ms", "planes_ms", "term_planes_ms", "create_ms", "create_plan_ms")
TOT_KEYS = ("matches", "algorithmic_bytes", "dense_algorithmic_bytes", "cand_algorithmic_bytes", "fused_algorithmic_bytes", "dense_queries", "pset_queries", "pset_algorithmic_bytes", "pset_bound_bytes", "probe_queries", "probe_algorithmic_bytes", "probe_bound_bytes", "cand_queries", "fused_queries",
            "cand_needed_bytes", "phrase_algorithmic_bytes", "phrase_queries", "planes_algorithmic_bytes", "planes_queries", "plane_terms", "plane_bytes", "term_planes_decoded_bytes",
            "bound_bytes", "dense_bound_bytes", "cand_bound_bytes", "fused_bound_bytes", "planes_bound_bytes", "phrase_bound_bytes")  # fmt: skip


class DryDevice:
    """--dry-run: option store in place of a device handle."""

    def __init__(self):
        self.opts = {}

    def set_option(self, k, v):
        self.opts[k] = int(v)

    def memory(self):
        return {"pool_in_use_bytes": 0, "pool_idle_bytes": 0, "pinned_idle_bytes": 0, "device_free_bytes": 0, "device_total_bytes": 0}

    def close(self):
        pass


class DryBatch:
    """--dry-run: a batch planned by the REAL host planner (csrc/planner.hpp through libtrinity_host.so) that never runs: its result blocks
    are zeros with the shapes of tri_batch_counts_device / tri_batch_topk_device, on CPU tensors."""

    def __init__(self, hix, flags, topk, flat, opts):
        import torch

        from trinity_amd import hostplan as HP

        self.flags, self.topk, self.nq = flags, topk, flat[1].shape[0]
        self.plan = HP.HostPlan(hix.h, None, flags, topk, threads=4, options={k: v for k, v in opts.items() if k != "account_needed_bytes"}, flat=flat)
        self._blocks = {"counts": torch.zeros(self.nq, dtype=torch.int64)}
        if (flags & 2) and topk:
            self._blocks.update(docs=torch.zeros((self.nq, topk), dtype=torch.int32), scores=torch.zeros((self.nq, topk), dtype=torch.float32),
                                topk_counts=torch.zeros(self.nq, dtype=torch.int32))  # fmt: skip

    def run(self):
        pass

    def sync(self):
        pass

    def blocks(self):
        return self._blocks

    def info(self):
        d = dict.fromkeys(MS_KEYS + TOT_KEYS, 0.0)
        d.update(create_plan_ms=float(self.plan.ms.sum()), create_ms=float(self.plan.ms.sum()), last_run_ms=1e-6)
        d.update({k: float(self.plan.s[k]) for k in ("dense_queries", "cand_queries", "fused_queries", "planes_queries")})
        return d

    def counts(self):
        return self._blocks["counts"].numpy().astype("uint64")

    def topk_results(self):
        b = self._blocks
        return b["docs"].numpy().view("uint32"), b["scores"].numpy(), b["topk_counts"].numpy().view("uint32")

    def close(self):
        if self.plan:
            self.plan.close()
            self.plan = None


class DryIndex:
    def __init__(self, dev, seg):
        from trinity_amd import hostplan as HP

        self.dev, self.h, self.seg = dev, HP.HostIndex.from_segment(seg), seg

    @classmethod
    def from_segment(cls, dev, seg):
        return cls(dev, seg)

    def info(self):
        return {"index_bytes": int(self.seg.index.size), "postings": int(self.seg.sum_terms_docs)}

    def close(self):
        self.h.close()


class DryEngine:
    """--dry-run: the names Workload uses of the trinity_amd package, backed by the host planner alone."""

    def __init__(self, T):
        self.Segment, self.engine, self.Index = T.Segment, T.engine, DryIndex

    def Batch(self, ix, programs, flags, topk=0, flat=None):
        return DryBatch(ix, flags, topk, flat, ix.dev.opts)


class Workload:
    """One rank's shard of a SURVEY §8(d) workload on a device: the segments' indexes and, per engine batch of a step, the query programs
    in the flat form the C-ABI takes (prepared once: what a C++ caller hands to tri_batch_create)."""

    def __init__(self, T, W, dev, name, docs, vocab, total_queries, rank, world, segs, ixs, seed=1337):
        self.T, self.dev, self.name = T, dev, name
        self.parts, self.desc = W.build_parts(name, docs, vocab, 10, 42, total_queries, seed=seed)
        self.build_s = self.upload_s = 0.0
        for pt in self.parts:
            if pt.codec not in segs:
                t0 = time.time()
                segs[pt.codec] = T.Segment(docs, vocab, 10, 42, codec=pt.codec)
                self.build_s += time.time() - t0
                t0 = time.time()
                ixs[pt.codec] = T.Index.from_segment(dev, segs[pt.codec])
                self.upload_s += time.time() - t0  # one-time: format walk + directory / delta-stream / cell-index build on the host, then PCIe
        self.segs, self.ixs = segs, ixs
        self.progs = [pt.programs[rank::world] for pt in self.parts]  # interleaved shard: same mix on every rank
        self.flat = [T.engine.flatten(p) for p in self.progs]
        self.nq = sum(len(p) for p in self.progs)

    def create_set(self):
        return [self.T.Batch(self.ixs[pt.codec], None, pt.flags, topk=pt.topk, flat=fl) for pt, fl in zip(self.parts, self.flat)]


class RotatingWorkload:
    """A query STREAM: create_set() hands out the sets of several workloads (same shape, different query seeds) in turn, so that no two
    consecutive steps run the same programs (the headline loop replays one set: its head terms' planes, L2 and the Infinity Cache are warm
    by construction)."""

    def __init__(self, wls):
        import threading

        self.wls, self.i, self.lock = wls, 0, threading.Lock()
        self.parts, self.nq, self.desc = wls[0].parts, wls[0].nq, wls[0].desc

    def create_set(self):
        with self.lock:  # (two compiler threads draw from the stream)
            wl = self.wls[self.i % len(self.wls)]
            self.i += 1
        return wl.create_set()


def read_back(T, bs):  # what every caller needs on the host: match counts and, scored, the top-K blocks
    for b_ in bs:
        b_.counts()
        if (b_.flags & T.FLAG_ACCUMULATED_SCORE) and b_.topk:
            b_.topk_results()


class Pipeline:
    """create -> run -> sync -> read back [-> gather] with two sets of batches in flight and the compiler on a thread of its own: a step QUEUES the
    set the compiler has ready behind the running one (the engine stream never drains: with one set in flight the GPU idled about a
    quarter of a millisecond per step between one set's sync / read-back and the next one's first launch — cfg2: 1.66 ms per step around 1.44 ms
    of kernels) while the compiling thread works on the set after it (tri_batch_create: host planning + the plan's H2D copy; include/trinity_hip.h: one
    thread may compile while another runs / awaits / releases other batches of the same device), then awaits and reads back the older set.
    Between the barriers of a timed region of K steps exactly K sets are launched and run to completion: the set running at the region's start
    was awaited by the barrier before it, the last one launched is awaited by the barrier after it."""

    COMPILERS = 2  # host threads that compile sets side by side (tri_dev keeps as many planner contexts: include/trinity_hip.h)

    def __init__(self, T, wl, gathers=None, blocks_of=None, sync_stream=True, compilers=None):
        import queue
        import threading

        self.T, self.wl, self.gathers, self.blocks_of, self.sync_stream = T, wl, gathers, blocks_of, sync_stream
        self.cur = wl.create_set()  # on the engine stream (running or complete)
        # two compilers where a set is small and its step short (16 K queries: a create takes about as long as the step); ONE where a set's output regions
        # weigh gigabytes (the 100 K mixed batch: 20 GB a set, bound-allocated — every compiler in the loop is one more set alive, and its creates are a
        # tenth of the step anyway)
        set_bytes = sum(4 * int(b.info().get("out_capacity", 0)) for b in self.cur)
        self.ncompilers = compilers or (Pipeline.COMPILERS if set_bytes < (8 << 30) else 1)
        for b in self.cur:
            b.run()
        # up to five sets are alive in the loop (read back and not yet released, running, launched, compiled and waiting, being compiled — the fifth
        # only when the compiler starts a set before the main loop has released the one it read back: rare while a create is shorter than a
        # step, the rule once they take about as long).  Their buffers go into the device pool NOW, so that no timed step pays a cold
        # allocation (cfg2: a 6 GB output region, 77 ms in the middle of a timed region; cfg5: 17 GB, 0.2 .. 484 ms)
        spares = []
        try:
            for _ in range(3 + self.ncompilers):  # (one more set in a compiler's hand per compiler thread)
                spares.append(wl.create_set())
        except Exception:  # (a batch so large that five sets do not fit the device: the loop then allocates what it needs as it goes — the pool gives idle buffers back)
            pass
        for spare in spares:
            for b in spare:
                b.close()
        # the compiler: tri_batch_create back to back on its own thread, one compiled set waiting at most (with a create per step handed over
        # by the main loop its 0.2 ms between a sync's return and the next hand-over added to every create: cfg2 1.69 ms per step around
        # 1.45 ms creates and 1.45 ms of kernels)
        self.ready = queue.Queue(maxsize=1)
        self.stop = False
        self.creates = []  # (start, end) of every create_set() of the compiler thread, time.perf_counter()

        def compile_loop():
            while not self.stop:
                try:
                    t_c = time.perf_counter()
                    bs = wl.create_set()
                    self.creates.append((t_c, time.perf_counter()))
                except BaseException as e:  # (handed to the main loop: a failed create fails the run)
                    self.ready.put(e)
                    return
                while not self.stop:
                    try:
                        self.ready.put(bs, timeout=0.05)
                        bs = None
                        break
                    except queue.Full:
                        pass
                if bs is not None:  # (stopped with a compiled set in hand)
                    for b in bs:
                        b.close()

        # ncompilers of them: a create is 0.6 - 1.1 ms of host planning for 16 K queries and does not scale past 8 - 16 host threads, so a loop whose step
        # is shorter than a create compiles two sets side by side (the device handle has two planner contexts); whichever is ready first is launched
        # next — a step still launches exactly one set and awaits exactly one
        self.compilers = [threading.Thread(target=compile_loop, daemon=True) for _ in range(self.ncompilers)]
        for th in self.compilers:
            th.start()
        self.done = None  # the last completed set: its results stay readable
        self.readback_s = 0.0
        self.gather_s = 0.0

    def step(self):
        T = self.T
        if self.done:  # (its buffers go back to the device pool)
            for b in self.done:
                b.close()
            self.done = None
        launched = self.ready.get()
        if isinstance(launched, BaseException):
            raise launched
        for b in launched:
            b.run()  # behind `cur` on the engine stream
        for b in self.cur:
            b.sync()
        t0 = time.perf_counter()
        read_back(T, self.cur)
        self.readback_s += time.perf_counter() - t0
        if self.gathers:  # result exchange straight from the engine's device buffers (docsets stay sharded in HBM)
            t0 = time.perf_counter()
            for g, b in zip(self.gathers, self.cur):
                g.rebind(self.blocks_of(b))
                g.step()
            if self.sync_stream:
                import torch

                torch.cuda.current_stream().synchronize()  # the receive side is complete before the send buffers go back to the pool
            self.gather_s += time.perf_counter() - t0
        infos = [b.info() for b in self.cur]
        for i, c in zip(infos, launched):  # the tri_batch_create calls of the set launched in this step (one set is compiled per step in the steady state)
            ci = c.info()
            i["create_ms"], i["create_plan_ms"] = ci["create_ms"], ci["create_plan_ms"]
        self.done, self.cur = self.cur, launched
        return infos

    def close(self):
        import queue

        self.stop = True
        for th in self.compilers:
            th.join()
        lefts = []
        while True:
            try:
                lefts.append(self.ready.get_nowait())
            except queue.Empty:
                break
        for bs in [self.cur, self.done] + [x for x in lefts if not isinstance(x, BaseException)]:
            for b in bs or []:
                b.close()
        self.cur = self.done = None


def timed(pipe, steps, warmup, barrier):
    for _ in range(warmup):
        pipe.step()
    barrier()
    pipe.readback_s = 0.0
    pipe.gather_s = 0.0
    acc = {}
    walls = []
    # (the cyclic collector stays out of the timed region: a generation-2 pass over the workloads' million-object program lists took milliseconds of a
    #  1.2 ms step now and then; nothing in the loop makes reference cycles)
    gc_was = gc.isenabled()
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    tp = t0
    for _ in range(steps):
        for i in pipe.step():
            for k in MS_KEYS:
                acc[k] = acc.get(k, 0.0) + i[k]
        tn = time.perf_counter()
        walls.append((tn - tp) * 1e3)
        tp = tn
    barrier()
    t1 = time.perf_counter()
    if gc_was:
        gc.enable()
    acc["_region"] = (t0, t1)
    acc["_walls"] = walls
    return t1 - t0, acc


def create_stats(pipe, region, steps, ms_per_step):
    """The tri_batch_create calls that RAN inside the timed region (the compiler thread's own clock around create_set(): all batches of a step),
    and whether the claim "planning is inside the loop" holds: the main loop takes one compiled set per step and the compiler holds at most two
    ahead (one queued, one in hand), so a region of K steps must see at least K - 2 creates END inside it (K - 1 - c with c compilers); and a compiler that needs longer per
    set than a step lasts would have to have been the loop's bound — the step time cannot be below the median create."""
    t0, t1 = region
    inside = sorted((e - s) * 1e3 for s, e in pipe.creates if t0 <= e <= t1)
    n = len(inside)
    med = inside[n // 2] if n else None
    nc = pipe.ncompilers
    ok = n >= steps - 1 - nc and (med is None or med <= nc * ms_per_step * 1.10)
    return {"creates_in_timed_region": n, "create_wall_ms": {"min": inside[0], "median": med, "max": inside[-1]} if n else None,
            "compiler_threads": nc, "create_ms_per_set": (med / nc) if med is not None else None,
            "planning_included": bool(ok),
            "what": "tri_batch_create calls (all batches of a step) that ran to their end inside the timed region, by the compiling thread's clock; compiler_threads of "
                    "them compile side by side (create_ms_per_set = median / compiler_threads: the rate sets become ready at); planning_included: at least "
                    "steps - 1 - compiler_threads creates ended inside the region (one set is queued, one in each compiler's hand at most) and their median does "
                    "not exceed compiler_threads x the step time"}  # fmt: skip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=0, help="queries per GPU per step (default: 16384; cfg3: 8192; cfg5: 12500 = a 100K batch over 8 GPUs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline sample budget (0 = skip)")
    ap.add_argument("--workload", default=None, choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"],
                    help="SURVEY §8(d) query sets; default: cfg2 at one GPU (the configuration BASELINE.json's metric is quoted on), cfg5 (the mixed 100K batch) at N > 1")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"], help="weak: --queries per GPU; strong: --queries in all, split over the GPUs.  Default: weak at N = 1 (the cfg2 "
                    "headline), STRONG at N > 1 — north_star's target is ONE 100 K-query mixed batch sharded over the node's GPUs")
    ap.add_argument("--scaling-total", type=int, default=SCALING_TOTAL_QUERIES, help="N = 1: queries of the mixed batch the scaling point's leg runs (north_star: 100000)")
    ap.add_argument("--scaling-ref-steps", type=int, default=3, help="steps of the scaling reference leg (0 = skip): N = 1: one rank's cfg5 shard on this GPU; N > 1: rank 0 alone on its shard")
    ap.add_argument("--rotating-sets", type=int, default=8, help="N = 1: distinct query sets (seeds 1337 ...) cycled through the loop in the rotating legs (0 / 1 = skip them)")
    ap.add_argument("--rotating-steps", type=int, default=16, help="timed steps of each rotating leg")
    ap.add_argument("--delivered-steps", type=int, default=3, help="N = 1: steps of the leg that also brings every docID set to pinned host memory (0 = skip)")
    ap.add_argument("--compilers", type=int, default=2, help="host threads that compile sets side by side in the loop (the device handle has two planner contexts)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="planner option (tri_dev_set_option), e.g. fused=0")
    ap.add_argument("--dry-run", action="store_true", help="launcher / sharding / gather plumbing WITHOUT a GPU (tests/test_bench_launch.py): the batches are planned by the real host "
                                                          "planner, nothing runs, the result blocks are zeros on CPU tensors gathered over gloo; the line says dry_run and measures nothing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...")
        args.gpus = world
    if args.workload is None:
        args.workload = "cfg2" if world == 1 else "cfg5"
    if args.scaling is None:
        args.scaling = "weak" if world == 1 else "strong"
    per_gpu_default = {"cfg3": 8192, "cfg5": 12500, "cfg1": 2000}.get(args.workload, 16384)
    if not args.queries:
        args.queries = per_gpu_default * (8 if args.scaling == "strong" else 1)  # strong: the 8-GPU batch (cfg5: 100K queries) at every N
    total_queries = args.queries * world if args.scaling == "weak" else args.queries

    import numpy as np
    import torch

    import trinity_amd as T
    from trinity_amd import dist as TD
    from trinity_amd import workloads as W

    dist = None
    dry = args.dry_run
    if not dry:
        torch.cuda.set_device(local_rank)
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:  # launched by torch.distributed.run: the result gather runs even with one rank
        import torch.distributed as dist

        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # one rank makes sure the native libraries are built (normally a no-op: the built .so travel with the tree); the
        # others wait instead of racing hipcc on the same output file
        if local_rank == 0:
            T.build.build_host() if dry else T.build_all()
        dist.barrier()
    else:
        T.build.build_host() if dry else T.build_all()

    def device_sync():
        if not dry:
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        device_sync()

    # ---- synthetic segments (identical on every rank) and this rank's query shard
    docs, vocab = (100_000, 10_000) if args.workload == "cfg1" and args.docs == 10_000_000 else (args.docs, args.vocab)
    dev = DryDevice() if dry else T.Device(local_rank)
    for o in args.option:
        k, v = o.split("=", 1)
        dev.set_option(k, int(v))
    segs, ixs = {}, {}
    wl = Workload(DryEngine(T) if dry else T, W, dev, args.workload, docs, vocab, total_queries, rank, world, segs, ixs)
    parts, nq_rank = wl.parts, wl.nq

    # ---- the diagnostic set: the same batches created once with account_needed_bytes (a directory walk per candidate-tile query and a
    #      pass over the plan: untimed) — its run gives the byte counts of the roofline block, its results are what the parity check,
    #      the CPU leg and the gather check read
    dev.set_option("account_needed_bytes", 1)
    batches = wl.create_set()
    dev.set_option("account_needed_bytes", 0)
    for b in batches:
        b.run()
    for b in batches:
        b.sync()
    infos = [b.info() for b in batches]
    tot = {k: float(sum(i[k] for i in infos)) for k in TOT_KEYS}

    dev_t = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    blocks_of = (lambda b: b.blocks()) if dry else (lambda b: TD.device_blocks(b, dev_t))
    gathers = [TD.ResultGather(dist, blocks_of(b)) for b in batches] if dist is not None else None
    Pipeline.COMPILERS = max(1, args.compilers)
    pipe = Pipeline(T, wl, gathers, blocks_of, sync_stream=not dry)

    # ---- the scaling curve's N = 1 point, measured in THIS run (before the timed region; every rank takes part so that the ranks stay in step).  The curve
    #      north_star asks for is ONE mixed 100 K-query batch (cfg5) sharded over the node's GPUs: strong scaling.  N > 1: rank 0 alone runs the WHOLE
    #      batch of this run (the other ranks parked at a barrier) — the same workload on one GPU; N = 1 (the cfg2 headline): this GPU runs the whole
    #      100 K mixed batch.  Either way the line carries `scaling_point` with the same-workload one-GPU rate beside its own.
    n1 = None
    if args.scaling_ref_steps > 0:
        n1_total = total_queries if world > 1 else args.scaling_total
        n1_workload = args.workload if world > 1 else "cfg5"
        if world > 1 or (args.workload != "cfg5" and docs == 10_000_000) or dry:
            barrier()
            if rank == 0:
                t0 = time.time()
                n1_wl = wl if (world == 1 and n1_workload == args.workload and n1_total == total_queries) else \
                    Workload(DryEngine(T) if dry else T, W, dev, n1_workload, docs, vocab, n1_total, 0, 1, segs, ixs)
                solo = Pipeline(T, n1_wl, compilers=1)  # (a 100 K batch plans in a few ms against a step of tens: one compiler, one set fewer alive)
                # (three warm-up steps: the sets alive in the loop each allocate their output regions once — up to half a second for a 17 GB region;
                #  from the fourth create on the device pool recycles them)
                el, _ = timed(solo, args.scaling_ref_steps, 3, device_sync)
                mem = dev.memory()
                solo.close()
                n1 = {"workload": n1_wl.desc, "scaling": "strong", "total_queries": n1_wl.nq, "n_gpus": 1, "steps": args.scaling_ref_steps, "value": n1_wl.nq * args.scaling_ref_steps / el,
                      "unit": "queries/s", "ms_per_step": el * 1e3 / args.scaling_ref_steps, "hbm_bytes_in_use": hbm_in_use(mem), "setup_s": time.time() - t0,
                      "what": "ONE GPU (rank 0, the other ranks parked at a barrier) on the whole batch: the same create -> run -> sync -> read-back loop, no gather"}  # fmt: skip
            barrier()

    # ---- the timed region
    elapsed, acc = timed(pipe, args.steps, args.warmup, barrier)
    readback_ms = pipe.readback_s * 1e3 / max(1, args.steps)
    gather_ms = pipe.gather_s * 1e3 / max(1, args.steps)
    mem_after = dev.memory()
    region = acc.pop("_region")
    walls = sorted(acc.pop("_walls"))
    cstats = create_stats(pipe, region, args.steps, elapsed * 1e3 / max(1, args.steps))

    # ---- legs that make the headline harder to flatter (N = 1, after the timed region; none of them feeds `value`)
    rotating = delivered = None
    if world == 1 and not dry and args.rotating_sets > 1 and args.rotating_steps > 0:
        # (b) a query STREAM: args.rotating_sets distinct sets (query seeds 1337, 1338, ...) cycled through the same loop.  `warm`: every head term's
        #     plane row is in the index's cache (a cycle of warm-up built them).  `cold_planes`: option planes_rebuild — every run decodes the plane
        #     rows its batch names again, i.e. what the stream pays when each step's head terms have just been evicted (k_term_planes inside the step)
        wls = [wl] + [Workload(T, W, dev, args.workload, docs, vocab, total_queries, rank, world, segs, ixs, seed=1337 + i) for i in range(1, args.rotating_sets)]
        legs = {}
        for leg, rebuild in (("warm", 0), ("cold_planes", 1)):
            dev.set_option("planes_rebuild", rebuild)
            rp = Pipeline(T, RotatingWorkload(wls))
            # (two cycles of warm-up: the sets' output regions differ in size, and the device pool must have seen every size that can be alive at once —
            #  with one cycle a timed create still met a cold 1.6 GB hipMalloc now and then: 76 ms in a 1.4 ms step)
            el, racc = timed(rp, args.rotating_steps, 2 * args.rotating_sets + 2, device_sync)
            rreg = racc.pop("_region")
            racc.pop("_walls")
            legs[leg] = {"value": nq_rank * args.rotating_steps / el, "ms_per_step": el * 1e3 / args.rotating_steps, "kernel_ms_per_step": racc["last_run_ms"] / args.rotating_steps,
                         "term_planes_ms_per_step": racc.get("term_planes_ms", 0.0) / args.rotating_steps, **{k: v for k, v in create_stats(rp, rreg, args.rotating_steps, el * 1e3 / args.rotating_steps).items() if k != "what"}}  # fmt: skip
            rp.close()
        dev.set_option("planes_rebuild", 0)
        rotating = {"sets": args.rotating_sets, "steps": args.rotating_steps, "unit": "queries/s", **legs,
                    "what": "the same create -> run -> sync -> read-back loop over a STREAM of distinct query sets (seeds 1337 ...; the headline loop replays ONE set): `warm` with every "
                            "head term's plane row cached with the index, `cold_planes` with the rows a step names decoded again in that step (planes_rebuild: every row evicted)"}  # fmt: skip
    if world == 1 and not dry and args.delivered_steps > 0:
        # (c) delivery: the docID sets themselves brought to the host — what MatchedIndexDocumentsFilter::consider(ids, cnt) (matches.h:161-165) is fed — into PINNED
        #     memory, one device-side gather + one copy per batch.  Two legs: `as_docids` — tri_batch_docsets, every set as ascending 4-byte docIDs (dense results
        #     expanded on the device first), serial: create -> run -> sync -> deliver; and the leg `value` is taken from — tri_batch_docsets_mixed, every set in the
        #     form the engine holds it (a dense set crosses PCIe as the words of its bitmap), OVERLAPPED: set n is delivered on the read-back stream while set n + 1's
        #     kernels run (the delivery waits outside the handle's lock)
        docs_parts = [i for i, pt in enumerate(parts) if not (pt.flags & T.FLAG_ACCUMULATED_SCORE)]
        need = [int(batches[i].counts().sum()) for i in docs_parts]
        if docs_parts and max(need) * 4 <= (6 << 30):
            pinned = torch.empty(max(need) + 64, dtype=torch.int32).pin_memory()

            def deliver(bs, mixed):
                nbytes = 0
                for i, b in enumerate(bs):
                    b.sync()
                    if i in docs_parts:
                        offs = (b.docsets_mixed(out=pinned) if mixed else b.docsets(out=pinned))[1]
                        nbytes += int(offs[-1]) * 4
                    else:
                        read_back(T, [b])
                return nbytes

            bytes_step = 0
            for it in range(args.delivered_steps + 1):  # (the first one untimed)
                if it == 1:
                    device_sync()
                    t0 = time.perf_counter()
                bs = wl.create_set()
                for b in bs:
                    b.run()
                bytes_step = deliver(bs, False)
                for b in bs:
                    b.close()
            el = time.perf_counter() - t0
            as_docids = {"value": nq_rank * args.delivered_steps / el, "unit": "queries/s", "ms_per_step": el * 1e3 / args.delivered_steps, "docid_bytes_per_step": bytes_step,
                         "host_GBps": bytes_step * args.delivered_steps / el / 1e9}  # fmt: skip
            prev = wl.create_set()
            for b in prev:
                b.run()
            msteps = 2 * args.delivered_steps
            for it in range(msteps + 1):  # (the first one untimed)
                if it == 1:
                    device_sync()
                    t0 = time.perf_counter()
                nxt = wl.create_set()
                for b in nxt:
                    b.run()  # behind `prev` on the engine stream
                bytes_step = deliver(prev, True)  # ... whose delivery (read-back stream) runs beside these kernels
                for b in prev:
                    b.close()
                prev = nxt
            for b in prev:
                b.sync()
            el = time.perf_counter() - t0
            for b in prev:
                b.close()
            delivered = {"value": nq_rank * msteps / el, "unit": "queries/s", "ms_per_step": el * 1e3 / msteps, "steps": msteps,
                         "docid_bytes_per_step": bytes_step, "host_GBps": bytes_step * msteps / el / 1e9, "as_docids": as_docids,
                         "what": "create -> run -> sync -> tri_batch_docsets_mixed: EVERY query's docID set in the form the engine holds it — ascending docIDs, or the words of a "
                                 "bitmap over the docID range for the dense ones — gathered on the device and copied into pinned host memory (the feed of consider(ids, cnt), which "
                                 "expands a bitmap on its side), set n delivered while set n + 1 runs; PCIe-bound — `value` above leaves the sets in HBM.  as_docids: every set as "
                                 "4-byte docIDs (tri_batch_docsets), serial"}  # fmt: skip
            del pinned
        else:
            delivered = {"skipped": "no DocumentsOnly part" if not docs_parts else f"{max(need) * 4 / 2**30:.1f} GiB of docIDs per step: not brought to the host in this leg"}

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        m = torch.tensor([tot["matches"], tot["algorithmic_bytes"]], dtype=torch.float64, device=dev_t)
        dist.all_reduce(m, op=dist.ReduceOp.SUM)
        matches_all, alg_all = float(m[0].item()), float(m[1].item())
    else:
        matches_all, alg_all = tot["matches"], tot["algorithmic_bytes"]

    gather_check = None
    if dist is not None:
        # what arrived in this rank's slot of every gathered block is what the engine reports locally (host copies through the C-ABI)
        ok = True
        for b, g in zip(pipe.done, gathers):
            ok &= bool(np.array_equal(g.recv["counts"][rank].cpu().numpy().astype(np.uint64), b.counts()))
            if "docs" in g.recv:
                d, s_, c = b.topk_results()
                ok &= bool(np.array_equal(g.recv["docs"][rank].cpu().numpy().view(np.uint32), d) and np.array_equal(g.recv["scores"][rank].cpu().numpy(), s_)
                           and np.array_equal(g.recv["topk_counts"][rank].cpu().numpy().view(np.uint32), c))  # fmt: skip
        # ... and the pipelined batches answer what the diagnostic set answered
        for b, b0 in zip(pipe.done, batches):
            ok &= bool(np.array_equal(b.counts(), b0.counts()))
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        gather_check = {"ranks": world, "blocks": sorted({k for g in gathers for k in g.recv}), "equal_on_every_rank": bool(t.item() == 1.0)}
    # (c) how many of the step's matches exist as docIDs in HBM (4 bytes each) and how many as bits of a result bitmap (RESULT_BITMAP: dense results)
    mat_ids = mat_bits = bitmap_queries = bitmap_bytes = 0
    for pt, b0 in zip(parts, batches):
        cnts = b0.counts()
        if (pt.flags & T.FLAG_ACCUMULATED_SCORE) and pt.topk:
            mat_ids += int(np.minimum(cnts, pt.topk).sum())  # (a scored top-K batch materialises its top-K lists)
            continue
        if dry:
            continue
        words = np.array([b0.docset_bitmap_words(q) for q in range(len(cnts))], dtype=np.int64)
        forms = words > 0
        mat_bits += int(cnts[forms].sum())
        mat_ids += int(cnts[~forms].sum())
        bitmap_queries += int(forms.sum())
        bitmap_bytes += int(words.sum()) * 4  # (a RESULT_BITMAP region is one bit per document of the query's docID range, whatever it matches)
    same_as_resident = all(bool(np.array_equal(b.counts(), b0.counts())) for b, b0 in zip(pipe.done, batches))

    if rank == 0:
        steps = max(1, args.steps)
        ms_per_step = elapsed * 1e3 / steps
        qps = nq_rank * world * steps / elapsed
        kms = {k: acc.get(KMS[k], 0.0) / steps for k in KERNELS}
        k_ms = acc["last_run_ms"] / steps  # HIP events around the whole run, per step (all batches of the step)
        rest_ms = acc["rest_ms"] / steps
        tp_ms = acc.get("term_planes_ms", 0.0) / steps
        dom = max(kms, key=kms.get)
        traffic, traffic_src = pmc_traffic(args, world)

        def gbs(b, ms):
            return b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0

        def frac(b, ms):
            return gbs(b, ms) / HBM_PEAK_GBS

        def kentry(k):
            tr = (traffic or {}).get(k)
            e = {"kernel_ms": kms[k], "queries": int(tot[KQ[k]]),
                 # the batch-level bound of the kernel's own queries: their distinct lists once + their output once
                 "bound_bytes_per_launch": tot[KBOUND[k]], "achieved": gbs(tot[KBOUND[k]], kms[k]), "frac": frac(tot[KBOUND[k]], kms[k]),
                 "traffic": tr, "physical_frac": frac(tr, kms[k]) if tr else None,
                 # sum over the kernel's queries of docbytes(t) + output — charges a shared list once per query that names it and
                 # counts lists a gallop skips: an effective rate, not a fraction of anything
                 "per_query_algorithmic": {"bytes_per_launch": tot[KALG[k]], "effective_GBps": gbs(tot[KALG[k]], kms[k])}}  # fmt: skip
            if k == "k_and" and tot["cand_needed_bytes"]:
                # what a perfect gallop must read for these queries, query by query (lead lists + the blocks that can hold a lead candidate + output)
                e["gallop_model_bytes_per_launch"] = tot["cand_needed_bytes"]  # (a MODEL of what a perfect gallop would read — k_and probes plane rows instead: these bytes are not moved, no fraction)
            return e

        # (the term planes live with the index: k_term_planes ran once, in the warm-up — the profile saw that one dispatch, a timed step has none)
        ONCE = ("k_term_planes", "k_term_plane0", "k_term_hits")  # once per index (the plane cache), not per step — unless the step rebuilt them (planes_rebuild)
        step_traffic = sum(v for k, v in (traffic or {}).items() if v and (k not in ONCE or tp_ms > 0.02)) if traffic else None
        info0 = ixs[parts[0].codec].info()
        if dry:
            k_ms = max(k_ms, 1e-9)
        out = {
            "metric": "queries/sec",
            "value": qps,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "step_wall_ms": {"min": walls[0], "median": walls[len(walls) // 2], "p90": walls[min(len(walls) - 1, len(walls) * 9 // 10)], "max": walls[-1]} if walls else None,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u32" if not any(pt.flags & T.FLAG_ACCUMULATED_SCORE for pt in parts) else "u32 docIDs / f64 sums of f32 BM25 terms",
            "data": "synthetic",
            **({"dry_run": "NO DEVICE: launcher / sharding / planner / gather plumbing only — value measures nothing"} if dry else {}),
            "config": {
                "workload": f"{wl.desc}, Zipf(1.0) {docs} docs / {vocab} terms",
                "docs": docs,
                "vocab": vocab,
                "queries_per_gpu_per_step": nq_rank,
                "queries_per_step": nq_rank * world,
                "batches_per_step": [{"part": pt.name, "queries": len(sp), "codec": "google" if pt.codec == T.engine.CODEC_GOOGLE else "lucene",
                                      "mode": "AccumulatedScore top-%d" % pt.topk if pt.flags & T.FLAG_ACCUMULATED_SCORE else "DocumentsOnly"} for pt, sp in zip(parts, wl.progs)],  # fmt: skip
                "index_bytes": int(info0["index_bytes"]),
                "postings": int(info0["postings"]),
                "parallelism": f"query-sharded x{world}, index replicated" + (", per-step RCCL all_gather of counts + top-K blocks" if world > 1 else ""),
                "options": args.option,
            },
            "step": "tri_batch_create (host planning + the plan's H2D copy) -> tri_batch_run -> tri_batch_sync -> match counts" + (" + top-K blocks" if any(pt.topk for pt in parts) else "") +
                    " read back to the host" + (" -> RCCL all_gather of the result blocks" if world > 1 else "") + "; two sets of batches in flight and the compilers on host threads of their own (end_to_end.compiler_threads): a step launches the set a compiler has ready behind the running one, then awaits the older one",
            "value_excludes": "the docID sets' way to the host: they stay in HBM, the host reads match counts / top-K blocks (PCIe-inclusive figure: DESIGN.md §5)",
            "per_gpu_value": qps / world,
            "matched_docids_per_sec": matches_all * steps / elapsed,
            "matches_per_step": matches_all,
            **({"docids_materialised_per_sec": mat_ids * steps / elapsed, "docids_materialised_per_step": mat_ids, "matches_kept_as_bitmap_bits_per_step": mat_bits, "bitmap_queries_per_step": bitmap_queries, "bitmap_bytes_per_step": bitmap_bytes,
                "materialised_what": "of this rank's matches per step: written to HBM as 4-byte docIDs (scored batches: their top-K lists) / kept as bits of a result bitmap "
                                     "(tri_batch_docset_bitmap; tri_batch_docsets expands them on delivery)"} if world == 1 else {}),
            "pipelined_results_equal_resident_batch": same_as_resident,
            "kernels_only": {"value": nq_rank * world / (k_ms * 1e-3) if k_ms > 0 else None, "ms_per_step": k_ms, "unit": "queries/s",
                             "what": "the step's kernels alone (HIP events around tri_batch_run inside the timed steps): no planning, no read-back"},
            "end_to_end": {"batch_create_ms": acc["create_ms"] / steps, "batch_create_plan_ms": acc["create_plan_ms"] / steps, "readback_ms": readback_ms, **cstats,
                           "batch_create_ms_what": "engine-reported tri_batch_create time of the sets LAUNCHED in the timed steps (they may have been compiled before the region began, and the time includes waiting for the device lock the running thread holds): see create_wall_ms for the creates that ran inside the region",
                           "what": "per step, inside the timed region: tri_batch_create (all batches of the step; its host-planner share) and the read-back of the match counts" +
                                   (" + top-K blocks" if any(pt.topk for pt in parts) else "")},
            "roofline": {
                "bound": "hbm",
                "scope": "step",
                # the batch-level bound: every DISTINCT list the step's queries name read once + every output written once
                "bound_bytes": tot["bound_bytes"],
                "achieved": gbs(tot["bound_bytes"], k_ms),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": frac(tot["bound_bytes"], k_ms),
                "kernel_ms": k_ms,
                "traffic": step_traffic,
                "traffic_source": traffic_src,
                "physical_frac": frac(step_traffic, k_ms) if step_traffic else None,  # PMC bytes / kernel time / peak (Infinity-Cache hits included)
                "dominant_kernel": dom,
                "kernels": {k: kentry(k) for k in KERNELS if tot[KQ[k]] > 0},
                "post_passes_ms": rest_ms,  # k_score (queries matched by k_and) / k_topk_merge / k_rich
                # the head terms the batch's queries share are decoded ONCE per launch (k_term_planes): its time is part of the step
                "term_planes": {"kernel_ms": tp_ms, "terms": int(tot["plane_terms"]), "decoded_list_bytes_per_launch": tot["term_planes_decoded_bytes"],
                                "scratch_bytes": tot["plane_bytes"], "traffic": sum((traffic or {}).get(k) or 0 for k in ("k_term_planes", "k_term_plane0")) or None},  # fmt: skip
                "per_query_algorithmic": {"bytes_per_step": tot["algorithmic_bytes"], "effective_GBps": gbs(tot["algorithmic_bytes"], k_ms),
                                          "what": "SURVEY §8(d): sum over queries of docbytes(t) + output; a list shared by n queries counts n times, skipped blocks count: no fraction"},
            },
            "segment_build_s": wl.build_s,
            "index_upload_s": wl.upload_s,
        }
        if rotating is not None:
            out["rotating"] = rotating
            out["value_rotating"] = rotating["cold_planes"]["value"]
        if delivered is not None:
            out["delivered"] = delivered
        out["hbm_bytes_in_use"] = hbm_in_use(mem_after)
        if world > 1:
            out["gather_ms"] = gather_ms
        # the scaling curve's point this line stands for, self-contained: the mixed batch, strong scaling, with the same workload's one-GPU rate beside it
        if world > 1:
            out["scaling_point"] = {"workload": wl.desc, "scaling": args.scaling, "total_queries": nq_rank * world, "n_gpus": world, "value": qps, "unit": "queries/s",
                                    "per_gpu_value": qps / world, "ms_per_step": ms_per_step, "gather_ms": gather_ms, "hbm_bytes_in_use": hbm_in_use(mem_after), "n1": n1,
                                    "speedup_vs_n1": qps / n1["value"] if n1 else None,
                                    "what": "this run: N ranks, the batch split over them, per-step RCCL all_gather of counts + top-K blocks; n1: rank 0 alone on the whole batch, same run"}  # fmt: skip
        elif n1 is not None:
            out["scaling_point"] = {**{k: v for k, v in n1.items() if k not in ("what", "setup_s")}, "per_gpu_value": n1["value"], "gather_ms": None, "n1": n1, "speedup_vs_n1": 1.0,
                                    "what": "the N = 1 point of the scaling curve (`value` above is the cfg2 headline, a different workload): this GPU on the whole mixed 100 K-query "
                                            "batch bench.py --gpus N > 1 splits over N GPUs"}  # fmt: skip
        if gather_check is not None:
            out["gather_check"] = gather_check
        if args.cpu_seconds > 0 and world == 1 and not dry:  # the CPU leg (and the per-query parity check that rides on it) runs at N = 1 only
            out["cpu_baseline"], out["parity_check"] = cpu_baseline(segs, parts, wl.progs, batches, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    pipe.close()
    for b in batches:
        b.close()
    for ix in ixs.values():
        ix.close()
    dev.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(args, world):
    """HBM bytes per launch of each kernel from the COMMITTED rocprofv3 PMC passes of this exact workload (FETCH_SIZE x2 per the gfx950
    correction + WRITE_SIZE; profiles/pmc_latest.json) — PMC counters cannot be read from inside the timed run, so the figures are
    quoted from that file (named in roofline.traffic_source) and only when docs / vocab / queries match."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            p = json.load(f)["bench_" + args.workload]
        if (p["docs"], p["vocab"], p["queries"]) == (args.docs, args.vocab, args.queries):
            from trinity_amd.build import kernels_stamp

            if p.get("kernels_stamp") != kernels_stamp():  # the counters were collected on other kernels than the ones this run times: not quoted
                return None, f"profiles/pmc_latest.json is STALE for this build (collected at kernels_stamp {p.get('kernels_stamp')}, git {p.get('git_head')}; this tree: {kernels_stamp()}): traffic not quoted"
            return {k: v["traffic_bytes_per_launch"] for k, v in p["kernels"].items()}, "profiles/pmc_latest.json (" + p.get("collected", "committed rocprofv3 --pmc passes") + f"; kernels_stamp {p['kernels_stamp']}, git {p.get('git_head')})"
    except Exception:
        pass
    return None, None


def program_text(prog):
    """A postfix program (include/trinity_hip.h: TRI_TOK) as the query text the reference's parser reads; None when it has no such text (matchsome's
    threshold is not part of the text)."""
    st = []
    for tok in [int(x) for x in prog]:
        op, arg = tok >> 28, tok & 0x0FFFFFFF
        if op == 0:
            st.append(f"t{arg}")
            continue
        n = 2 if op in (4, 5) else arg
        if op == 6 or n < 1 or n > len(st):
            return None
        kids, st = st[len(st) - n :], st[: len(st) - n]
        if op == 1:
            st.append("(" + " ".join(kids) + ")")
        elif op == 2:
            st.append("(" + " OR ".join(kids) + ")")
        elif op == 3:
            st.append('"' + " ".join(kids) + '"')
        elif op == 4:
            st.append("(" + kids[0] + " NOT " + kids[1] + ")")
        else:
            st.append("(" + kids[0] + " <" + kids[1] + ">)")
    if len(st) != 1:
        return None
    t = st[0]
    return t[1:-1] if t.startswith("(") and t.endswith(")") and t.count("(") == 1 else t


def cpu_reference(seg, pt, progs, gcounts, budget_s):
    """The GENUINE reference on the same sample: oracle/_ref/ref_driver (the reference's own sources compiled where they lie, oracle/Makefile) indexes
    the same synthetic corpus with the reference's encoder and runs exec_query over the queries' text, one thread, the clock around exec_query only.
    Google codec only (the Lucene side of the reference needs a library this image lacks).  Never fails the run: an error is reported in its place."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    try:
        if not os.path.exists(exe):
            return {"error": "oracle/_ref/ref_driver is not built (it is built where /root/reference exists: __graft_entry__.build())"}
        texts = [program_text(p) for p in progs]
        if any(t is None for t in texts):
            return {"error": "a sampled program has no query text"}
        scored = bool(pt.flags & 2)
        ncores = len(os.sched_getaffinity(0))
        inp = f"timed {2 if scored else 1} {budget_s:.3f} {len(texts)} {ncores}\n" + "\n".join(texts) + "\n"
        t0 = time.perf_counter()
        r = subprocess.run([exe, str(seg.D), str(seg.V), str(seg.slots), str(seg.seed)], input=inp, capture_output=True, text=True, timeout=budget_s * 4 + 240)
        wall = time.perf_counter() - t0
        lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        j = next(x for x in lines if x.get("cmd") == "timed")
        j.update(next((x for x in lines if x.get("cmd") == "timedmt"), {}))  # (a second line: the all-cores pass, when it ran to its end)
        n, secs = int(j["queries"]), float(j["seconds"])
        same = [int(c) for c in j["counts"]] == [int(c) for c in gcounts[:n]]
        all_cores = None
        if "mt_seconds" in j:  # (DocumentsOnly) the same queries, one per thread at a time, on every host core
            all_cores = {"value": n / float(j["mt_seconds"]), "unit": "queries/s", "cores": int(j["threads"]), "matched_docids_per_sec": int(j["matches"]) / float(j["mt_seconds"]),
                         "match_counts_equal_single_thread": bool(j["mt_counts_equal"]),
                         "sample": f"the same {n} queries in {float(j['mt_seconds']):.2f}s, exec_query on {j['threads']} threads (one query per thread at a time, shared read-only index source); "
                                   f"a sample this small is bound by its few heaviest queries, not by the core count"}  # fmt: skip
        return {"value": n / secs, "unit": "queries/s", "cores": 1, "kind": "reference", **({"all_cores": all_cores} if all_cores else {}),
                "sample": f"first {n} queries of rank 0's shard ({j['matches']} matches) in {secs:.1f}s: exec_query of the reference compiled from its own sources (oracle/_ref/ref_driver, "
                          f"`timed`), one thread, {'AccumulatedScoreScheme + BM25' if scored else 'DocumentsOnly'}; corpus + index built by the reference's encoder in {wall - secs:.1f}s (untimed)",
                "matched_docids_per_sec": int(j["matches"]) / secs, "host_cpus": os.cpu_count(), "match_counts_equal_gpu": same}  # fmt: skip
    except Exception as e:  # a reported baseline: the line stands without it
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def cpu_baseline(segs, parts, shard_progs, batches, budget_s, reference=True):
    """The CPU oracle (plain-C restatement of the reference's iterator path), one thread, on the first programs of every part of rank
    0's shard until the time budget is spent; a reported baseline only.  Every sampled query is also a full-size parity check:
    per-query match counts, FNV-1a of the docID set (DocumentsOnly) or the top-K docIDs (scored) must equal the GPU's.  For
    DocumentsOnly 2-term batches the all-cores figure (one query per thread) is added."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O

    n = matches = 0
    bad = []
    used = []  # per part: how many of its programs the port leg ran
    checked = {"counts": 0, "docset_hashes": 0, "topk_lists": 0}
    t_total = 0.0
    all_cores = None
    for pi, (pt, progs, batch) in enumerate(zip(parts, shard_progs, batches)):
        seg = segs[pt.codec]
        if seg.codec == 2:
            ora = O.Index.generate(seg.D, seg.V, seg.slots, seg.seed, codec="lucene")
        else:
            ora = O.Index.wrap(seg.index, seg.terms, seg.docs_cnt, seg.sum_terms_docs, seg.sum_term_hits)
        scored = bool(pt.flags & 2)
        gcounts = batch.counts()
        ghash = batch.docset_hashes() if not scored else None
        gtop = batch.topk_results() if scored and pt.topk else None
        share = budget_s * len(progs) / max(1, sum(len(p) for p in shard_progs))
        t0 = time.perf_counter()
        for qi, p in enumerate(progs):
            docs, scores = ora.exec(p, O.FLAG_ACCUM_SCORE if scored else O.FLAG_DOCUMENTS_ONLY)
            dt_q = time.perf_counter()
            matches += len(docs)
            n += 1
            if int(gcounts[qi]) != len(docs):
                bad.append((pi, qi, "count", int(gcounts[qi]), len(docs)))
            checked["counts"] += 1
            if ghash is not None:
                if int(ghash[qi]) != O.fnv1a_docs(docs):
                    bad.append((pi, qi, "docset hash"))
                checked["docset_hashes"] += 1
            if gtop is not None:
                td, ts = ora.topk(docs, scores, pt.topk)
                if gtop[0][qi, : len(td)].tolist() != td.tolist() or not np.allclose(gtop[1][qi, : len(td)], ts, rtol=1e-5, atol=0):
                    bad.append((pi, qi, "top-k"))
                checked["topk_lists"] += 1
            t_total -= time.perf_counter() - dt_q  # the checking is not part of the baseline
            if time.perf_counter() - t0 > share and qi + 1 >= 16:
                break
        t_total += time.perf_counter() - t0
        used.append(qi + 1)
        if pi == 0 and not scored and all(len(p) == 3 for p in progs[:8]):
            # also one query per thread on all host cores (the reference's exec_query is re-entrant per thread,
            # exec.cpp:12).  Same oracle, same queries, drawn heaviest first from a shared cursor by C threads.
            try:
                ncores = len(os.sched_getaffinity(0))
                pa = np.array([p for p in progs[: qi + 1] if len(p) == 3], dtype=np.uint32)  # the SAME sample the single-thread figure above was taken on
                done, m, dt2 = ora.exec_batch_mt(pa, O.FLAG_DOCUMENTS_ONLY, ncores, max(30.0, budget_s * 3))
                all_cores = {"value": done / dt2, "unit": "queries/s", "cores": ncores, "matched_docids_per_sec": m / dt2,
                             "sample": f"the same first {done} 2-term queries ({m} matches) in {dt2:.2f}s, one query per thread (pthreads, oracle to_exec_batch_mt); a sample this "
                                       f"small is bound by its few heaviest queries, not by the core count"}  # fmt: skip
            except Exception as e:  # the single-thread figure stands on its own
                all_cores = {"error": str(e)}
    res = {"value": n / t_total, "unit": "queries/s", "cores": 1, "kind": "port",
           "sample": f"first {n} programs of rank 0's shard, proportionally from every part ({matches} matches) in {t_total:.1f}s, oracle/trinity_oracle.c single thread",
           "matched_docids_per_sec": matches / t_total, "host_cpus": os.cpu_count()}  # fmt: skip
    if all_cores is not None:
        res["all_cores"] = all_cores
    parity = {"queries": n, "per_query": checked, "mismatches": len(bad), "first_mismatches": bad[:5], "equal": not bad}
    if reference and len(parts) == 1 and parts[0].codec == 1:
        # the same sample through the genuine reference (when its driver is there): that figure leads, the port's stands beside it
        ref = cpu_reference(segs[1], parts[0], shard_progs[0][: used[0]], batches[0].counts(), budget_s)
        if "value" in ref:
            parity["reference_match_counts_equal"] = ref.pop("match_counts_equal_gpu")
            ref["port"] = res
            res = ref
        else:
            res["reference"] = ref
    return res, parity


if __name__ == "__main__":
    main()
