#!/usr/bin/env python3
"""bench.py — the hot path's headline benchmark on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Default workload (BASELINE.json configs[1]): batched 2-term AND queries (Zipf-sampled terms, seed 1337,
distinct within a query) over the 10M-document / 1M-term synthetic Zipf(1.0) segment (corpus seed 42, google_codec),
DocumentsOnly mode — every query's full ascending docID set is materialised in HBM.  --workload cfg3 / cfg4 / cfg5 / cfg1 are
the other query sets of SURVEY §8(d) (cfg5 = the mixed batch: a DocumentsOnly batch on the google_codec segment plus a BM25
top-100 batch on the lucene_codec segment of the same corpus, both per step).
One "step" = one pass of the engine over one batch of --queries queries per GPU (index and compiled batches already resident
in HBM).  Multi-GPU: one process per GPU, the index replicated, the query stream sharded (queries are independent, exec.h:57-62):
--scaling weak (default) gives every rank a shard of --queries queries, --scaling strong splits --queries over the ranks; with
--gpus N > 1 the default workload is cfg5, the mixed 100K-query batch BASELINE.json's scaling criterion is quoted on (12500 queries
per GPU).  At the end of every step the ranks all_gather their result blocks over RCCL straight from the engine's device buffers:
per-query match counts and, for scored batches, the [Q/G][K] top-K docID/score blocks (trinity_amd/dist.py ResultGather — the
only exchange the path has).

Rank 0 prints ONE JSON line: metric/value = queries/s over all GPUs (kernels + result gather; the docID sets themselves stay in
HBM — copying every set back over PCIe would make the rate PCIe-bound); `roofline` = the dominant kernel's
algorithmic bytes (sum over its queries of docbytes(t) + 4 B per match, or + 8 B x min(matches, K) when scored)
/ its mean launch duration measured with HIP events on the engine's stream; `cpu_baseline` = the CPU oracle (restatement of the
reference exec path) timed on a bounded sample of the same batch, which doubles as a per-query full-size parity check.
`end_to_end` = what the kernel-only `value` leaves out: tri_batch_create (host planning + H2D) and the result read-back (match
counts, top-K blocks) per batch, and the rate of a create / run / read-back loop with the next batch compiled while the current
one runs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
KERNELS = ("k_and_dense", "k_and", "k_fused", "k_planes", "k_phrase")  # the kernels with per-launch HIP-event brackets and their own algorithmic bytes
KMS = {"k_and_dense": "dense_ms", "k_and": "cand_ms", "k_fused": "fused_ms", "k_planes": "planes_ms", "k_phrase": "phrase_ms"}
KALG = {"k_and_dense": "dense_algorithmic_bytes", "k_and": "cand_algorithmic_bytes", "k_fused": "fused_algorithmic_bytes", "k_planes": "planes_algorithmic_bytes",
        "k_phrase": "phrase_algorithmic_bytes"}  # fmt: skip
KQ = {"k_and_dense": "dense_queries", "k_and": "cand_queries", "k_fused": "fused_queries", "k_planes": "planes_queries", "k_phrase": "phrase_queries"}


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
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak: --queries per GPU; strong: --queries in all, split over the GPUs")
    ap.add_argument("--e2e-steps", type=int, default=3, help="create / run / read-back steps of the end-to-end leg (0 = skip)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="planner option (tri_dev_set_option), e.g. fused=0")
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
    torch.cuda.set_device(local_rank)
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:  # launched by torch.distributed.run: the result gather runs even with one rank
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # one rank makes sure the native libraries are built (normally a no-op: the built .so travel with the tree); the
        # others wait instead of racing hipcc on the same output file
        if local_rank == 0:
            T.build_all()
        dist.barrier()
    else:
        T.build_all()

    # ---- synthetic segments (identical on every rank) and this rank's query shard
    docs, vocab = (100_000, 10_000) if args.workload == "cfg1" and args.docs == 10_000_000 else (args.docs, args.vocab)
    parts, wl_desc = W.build_parts(args.workload, docs, vocab, 10, 42, total_queries)
    dev = T.Device(local_rank)
    dev.set_option("account_needed_bytes", 1)  # batch creation also works out what a perfect gallop must read for k_and's queries (untimed)
    for o in args.option:
        k, v = o.split("=", 1)
        dev.set_option(k, int(v))
    segs, ixs, build_s, upload_s = {}, {}, 0.0, 0.0
    for pt in parts:
        if pt.codec not in segs:
            t0 = time.time()
            segs[pt.codec] = T.Segment(docs, vocab, 10, 42, codec=pt.codec)
            build_s += time.time() - t0
            t0 = time.time()
            ixs[pt.codec] = T.Index.from_segment(dev, segs[pt.codec])
            upload_s += time.time() - t0  # one-time: format walk + directory / delta-stream / cell-index build on the host, then PCIe
    batches, shard_progs, shard_flat = [], [], []
    for pt in parts:
        mine = pt.programs[rank::world]  # interleaved shard: same mix on every rank
        shard_progs.append(mine)
        shard_flat.append(T.engine.flatten(mine))  # the (program words, tri_query table) pair the C-ABI takes: what a C++ caller hands over
        batches.append(T.Batch(ixs[pt.codec], None, pt.flags, topk=pt.topk, flat=shard_flat[-1]))
    nq_rank = sum(len(p) for p in shard_progs)

    def create_set():  # a step's batches compiled afresh (no needed-bytes accounting: that walk is a bench-only diagnostic)
        dev.set_option("account_needed_bytes", 0)
        out = []
        for pt, sp, fl in zip(parts, shard_progs, shard_flat):
            t_ = time.perf_counter()
            out.append(T.Batch(ixs[pt.codec], None, pt.flags, topk=pt.topk, flat=fl))
            if os.environ.get("BENCH_TRACE_CREATE"):
                print(f"[create] {pt.name}: {len(sp)} queries {(time.perf_counter() - t_) * 1e3:.2f} ms", file=sys.stderr, flush=True)
        dev.set_option("account_needed_bytes", 1)
        return out

    def read_back(bs):  # what every caller needs on the host: match counts and, scored, the top-K blocks
        for b_ in bs:
            b_.counts()
            if (b_.flags & T.FLAG_ACCUMULATED_SCORE) and b_.topk:
                b_.topk_results()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    gathers = []
    if dist is not None:
        gathers = [TD.ResultGather(dist, TD.device_blocks(b, torch.device("cuda", local_rank))) for b in batches]

    def step():
        for b in batches:
            b.run()
        for b in batches:
            b.sync()
        for g in gathers:  # result exchange straight from the engine's device buffers (docsets stay sharded in HBM)
            g.step()
        if gathers:
            torch.cuda.current_stream().synchronize()  # the receive side is complete before the next step rewrites the send buffers
        return [b.info() for b in batches]

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    acc = {}
    infos = None
    for _ in range(args.steps):
        infos = step()
        for i in infos:
            for k in ("last_run_ms", "dense_ms", "cand_ms", "fused_ms", "phrase_ms", "rest_ms", "planes_ms", "term_planes_ms"):
                acc[k] = acc.get(k, 0.0) + i[k]
    barrier()
    elapsed = time.perf_counter() - t0
    tot = {k: float(sum(i[k] for i in infos)) for k in ("matches", "algorithmic_bytes", "dense_algorithmic_bytes", "cand_algorithmic_bytes", "fused_algorithmic_bytes",
                                                       "dense_queries", "cand_queries", "fused_queries", "cand_needed_bytes", "phrase_algorithmic_bytes", "phrase_queries",
                                                       "planes_algorithmic_bytes", "planes_queries", "plane_terms", "plane_bytes", "term_planes_decoded_bytes")}  # fmt: skip

    # ---- end to end (rank 0's view; every rank runs it so the ranks stay in step): batch creation, read-back, and a create / run /
    #      read-back loop in which the next step's batches are compiled on the host while the current ones run on the device
    e2e = None
    if args.e2e_steps > 0:
        # (the device handle recycles the batches' large buffers — tri_dev's pool: two sets are in flight in the loop below, so two are
        #  created and released first, like the kernels' warm-up steps; a cold 15 GB hipMalloc was measured between 10 ms and 1 s)
        t1 = time.perf_counter()
        warm = [create_set(), create_set()]
        create_cold_ms = (time.perf_counter() - t1) * 1e3 / 2
        for ws in warm:
            for b_ in ws:
                b_.close()
        t1 = time.perf_counter()
        nxt = create_set()
        create_ms = (time.perf_counter() - t1) * 1e3
        t1 = time.perf_counter()
        read_back(batches)
        readback_ms = (time.perf_counter() - t1) * 1e3
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        cur = nxt
        for b_ in cur:
            b_.run()
        for _ in range(args.e2e_steps - 1):
            nxt = create_set()  # (host planning + H2D while `cur` runs on the engine stream)
            for b_ in cur:
                b_.sync()
            read_back(cur)
            for b_ in nxt:
                b_.run()
            for b_ in cur:
                b_.close()
            cur = nxt
        for b_ in cur:
            b_.sync()
        read_back(cur)
        loop_s = time.perf_counter() - t1
        for b_ in cur:
            b_.close()
        e2e = {"batch_create_ms": create_ms, "batch_create_cold_ms": create_cold_ms, "readback_ms": readback_ms, "readback": "match counts" + (" + top-K blocks" if any(pt.topk for pt in parts) else "") + " to the host (docID sets stay in HBM)",
               "steps": args.e2e_steps, "loop_ms_per_step": loop_s * 1e3 / args.e2e_steps,
               "queries_per_sec": nq_rank * world * args.e2e_steps / loop_s,
               "note": "create (host planning + H2D) -> run -> read-back per step, the next step's batches compiled while the current ones run; the first create is outside the loop; batch_create_cold_ms: before the device handle's buffer pool has anything to recycle"}  # fmt: skip
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        m = torch.tensor([tot["matches"], tot["algorithmic_bytes"]], dtype=torch.float64, device="cuda")
        dist.all_reduce(m, op=dist.ReduceOp.SUM)
        matches_all, alg_all = float(m[0].item()), float(m[1].item())
    else:
        matches_all, alg_all = tot["matches"], tot["algorithmic_bytes"]

    gather_check = None
    if dist is not None:
        # what arrived in this rank's slot of every gathered block is what the engine reports locally (host copies through the C-ABI)
        ok = True
        for b, g in zip(batches, gathers):
            ok &= bool(np.array_equal(g.recv["counts"][rank].cpu().numpy().astype(np.uint64), b.counts()))
            if "docs" in g.recv:
                d, s_, c = b.topk_results()
                ok &= bool(np.array_equal(g.recv["docs"][rank].cpu().numpy().view(np.uint32), d) and np.array_equal(g.recv["scores"][rank].cpu().numpy(), s_)
                           and np.array_equal(g.recv["topk_counts"][rank].cpu().numpy().view(np.uint32), c))  # fmt: skip
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        gather_check = {"ranks": world, "blocks": sorted({k for g in gathers for k in g.recv}), "equal_on_every_rank": bool(t.item() == 1.0)}

    if rank == 0:
        steps = max(1, args.steps)
        ms_per_step = elapsed * 1e3 / steps
        qps = nq_rank * world * steps / elapsed
        kms = {k: acc.get(KMS[k], 0.0) / steps for k in KERNELS}
        kalg = {k: tot[KALG[k]] for k in KERNELS}
        k_ms = acc["last_run_ms"] / steps
        rest_ms = acc["rest_ms"] / steps
        dom = max(kms, key=kms.get)

        def gbs(b, ms):
            return b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0

        def physical(k):  # PMC traffic of the kernel / its time / peak (None without a committed PMC pass of this exact workload)
            tr = (traffic or {}).get(k)
            return gbs(tr, kms[k]) / HBM_PEAK_GBS if tr and kms[k] > 0 else None

        def kentry(k):
            e = {"kernel_ms": kms[k], "algorithmic_bytes_per_launch": kalg[k], "achieved": gbs(kalg[k], kms[k]), "frac": gbs(kalg[k], kms[k]) / HBM_PEAK_GBS,
                 "queries": int(tot[KQ[k]]), "traffic": (traffic or {}).get(k), "physical_frac": physical(k)}  # fmt: skip
            if k == "k_and" and tot["cand_needed_bytes"]:
                # galloping skips, so algorithmic bytes are no bound for this kernel (its "achieved" can exceed the peak): the bound is what a
                # perfect gallop must read (tri_batch_info.cand_needed_bytes: lead lists + the blocks that can hold a lead candidate + output)
                e["needed_bytes_per_launch"] = tot["cand_needed_bytes"]
                e["needed_achieved"] = gbs(tot["cand_needed_bytes"], kms[k])
                e["needed_frac"] = e["needed_achieved"] / HBM_PEAK_GBS
            return e

        traffic, traffic_src = pmc_traffic(args, world)
        info0 = ixs[parts[0].codec].info()
        out = {
            "metric": "queries/sec",
            "value": qps,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u32" if not any(pt.flags & T.FLAG_ACCUMULATED_SCORE for pt in parts) else "u32 docIDs / f64 sums of f32 BM25 terms",
            "data": "synthetic",
            "config": {
                "workload": f"{wl_desc}, Zipf(1.0) {docs} docs / {vocab} terms",
                "docs": docs,
                "vocab": vocab,
                "queries_per_gpu_per_step": nq_rank,
                "queries_per_step": nq_rank * world,
                "batches_per_step": [{"part": pt.name, "queries": len(sp), "codec": "google" if pt.codec == T.engine.CODEC_GOOGLE else "lucene",
                                      "mode": "AccumulatedScore top-%d" % pt.topk if pt.flags & T.FLAG_ACCUMULATED_SCORE else "DocumentsOnly"} for pt, sp in zip(parts, shard_progs)],  # fmt: skip
                "index_bytes": int(info0["index_bytes"]),
                "postings": int(info0["postings"]),
                "parallelism": f"query-sharded x{world}, index replicated" + (", per-step RCCL all_gather of counts + top-K blocks" if world > 1 else ""),
                "options": args.option,
            },
            "matched_docids_per_sec": matches_all * steps / elapsed,
            "matches_per_step": matches_all,
            "value_excludes": "result delivery to the host: docID sets / top-K blocks stay in HBM (PCIe-inclusive figure: DESIGN.md §5)",
            "roofline": {
                "bound": "hbm",
                "achieved": gbs(kalg[dom], kms[dom]),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": gbs(kalg[dom], kms[dom]) / HBM_PEAK_GBS,
                "traffic": (traffic or {}).get(dom),
                "traffic_source": traffic_src,
                "physical_frac": physical(dom),  # PMC bytes / kernel time / peak: what actually crossed the fabric (algorithmic > physical where lists are
                                                 # skipped, shared between the batch's queries through the term planes, or served by the Infinity Cache)
                # ... and with the term planes' build charged to this kernel alone (it reads them instead of decoding the shared head terms per query)
                "frac_incl_term_planes": gbs(kalg[dom], kms[dom] + acc.get("term_planes_ms", 0.0) / steps) / HBM_PEAK_GBS,
                "kernel": dom,
                "kernel_ms": kms[dom],
                "algorithmic_bytes_per_launch": kalg[dom],
                "queries_per_launch": int(tot[KQ[dom]]),
                "other_kernels": {k: kentry(k) for k in KERNELS if k != dom and tot[KQ[k]] > 0},
                "post_passes_ms": rest_ms,  # k_score (queries matched by k_and) / k_topk_merge / k_rich: no bytes of their own
                # the head terms the batch's queries share are decoded ONCE per launch (k_term_planes) instead of once per query that names them:
                # its time is part of the step, its bytes are what it reads of the codec's lists
                "term_planes": {"kernel_ms": acc.get("term_planes_ms", 0.0) / steps, "terms": int(tot["plane_terms"]), "decoded_list_bytes_per_launch": tot["term_planes_decoded_bytes"],
                                "scratch_bytes": tot["plane_bytes"]},  # fmt: skip
                "whole_step": dict({"kernel_ms": k_ms, "algorithmic_bytes": tot["algorithmic_bytes"], "achieved": gbs(tot["algorithmic_bytes"], k_ms)},
                                   **({} if tot["cand_queries"] else {"frac": gbs(tot["algorithmic_bytes"], k_ms) / HBM_PEAK_GBS})),  # (no fraction when a skipping kernel is in the step)
            },
            "segment_build_s": build_s,
            "index_upload_s": upload_s,
        }
        if e2e is not None:
            out["end_to_end"] = e2e
        if gather_check is not None:
            out["gather_check"] = gather_check
        if args.cpu_seconds > 0 and world == 1:  # the CPU leg (and the per-query parity check that rides on it) runs at N = 1 only
            out["cpu_baseline"], out["parity_check"] = cpu_baseline(segs, parts, shard_progs, batches, args.cpu_seconds)
        print(json.dumps(out), flush=True)

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
            return {k: v["traffic_bytes_per_launch"] for k, v in p["kernels"].items()}, "profiles/pmc_latest.json (" + p.get("collected", "committed rocprofv3 --pmc passes") + ")"
    except Exception:
        pass
    return None, None


def cpu_baseline(segs, parts, shard_progs, batches, budget_s):
    """The CPU oracle (plain-C restatement of the reference's iterator path), one thread, on the first programs of every part of rank
    0's shard until the time budget is spent; a reported baseline only.  Every sampled query is also a full-size parity check:
    per-query match counts, FNV-1a of the docID set (DocumentsOnly) or the top-K docIDs (scored) must equal the GPU's.  For
    DocumentsOnly 2-term batches the all-cores figure (one query per thread) is added."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O

    n = matches = 0
    bad = []
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
    return res, parity


if __name__ == "__main__":
    main()
