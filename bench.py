#!/usr/bin/env python3
"""bench.py — the hot path's headline benchmark on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1]): batched 2-term AND queries (Zipf-sampled terms, seed 1337,
distinct within a query) over the 10M-document / 1M-term synthetic Zipf(1.0) segment (corpus seed 42,
google_codec), DocumentsOnly mode — every query's full ascending docID set is materialised in HBM.
One "step" = one pass of the engine over one batch of --queries queries per GPU (index and compiled batch
already resident in HBM).  Multi-GPU: one process per GPU, the index replicated, every rank runs its own
batch of the same size (weak scaling; queries are independent, exec.h:57-62), and the per-query match counts
are all-gathered over RCCL at the end of every step (the only exchange the DocumentsOnly path has).

Rank 0 prints ONE JSON line: metric/value = queries/s over all GPUs; `roofline` = algorithmic bytes (SURVEY §8d:
sum over queries of docbytes(t) of both terms + 4 B per match) / mean kernel time measured with HIP events on the
engine's stream; `cpu_baseline` = the CPU oracle (restatement of the reference exec path, one thread) timed on a
bounded sample of the same batch.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--vocab", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=16384, help="queries per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline sample budget (0 = skip)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"],
                    help="SURVEY §8(d) query sets; cfg2 (default) is the configuration BASELINE.json's metric is quoted on, the others are "
                         "extra measurements (whole-step roofline only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...")
        args.gpus = world

    import numpy as np
    import torch

    import trinity_amd as T
    from trinity_amd import dist as TD

    dist = None
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # one rank makes sure the native libraries are built (normally a no-op: the built .so travel with the tree); the
        # others wait instead of racing hipcc on the same output file
        if local_rank == 0:
            T.build_all()
        dist.barrier()
    else:
        torch.cuda.set_device(local_rank)
        T.build_all()

    # ---- synthetic segment (identical on every rank) and this rank's query batch
    t0 = time.time()
    from trinity_amd import workloads as W

    progs = None
    wl_desc = None
    codec = T.engine.CODEC_GOOGLE
    if args.workload != "cfg2":
        allp, wflags, wtopk, codec, wl_desc = W.build(args.workload, args.docs, args.vocab, 10, 42, args.queries * world)
        progs = allp[rank::world][: args.queries]  # interleaved shard: same mix on every rank
    seg = T.Segment(args.docs, args.vocab, 10, 42, codec=codec)
    build_s = time.time() - t0
    dev = T.Device(local_rank)
    t0 = time.time()
    ix = T.Index.from_segment(dev, seg)
    upload_s = time.time() - t0  # one-time: format walk + directory / delta-stream / cell-index build on the host, then PCIe
    info = ix.info()
    if progs is None:
        qall = T.gen_queries(args.vocab, 1337, args.queries * world, 2)
        qs = TD.shard_rows(qall, rank, world, args.queries)  # interleaved shard: same cost distribution on every rank
        batch = T.Batch.conjunctions(ix, qs, T.FLAG_DOCUMENTS_ONLY)
    else:
        qs = None
        batch = T.Batch(ix, progs, wflags, topk=wtopk)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    counts_dev = None
    if dist is not None:
        counts_dev = torch.zeros(args.queries, dtype=torch.int64, device="cuda")

    def step():
        batch.run()
        batch.sync()
        if dist is not None:
            # result exchange: per-query match counts to every rank (docsets stay sharded in HBM)
            counts_dev.copy_(torch.from_numpy(batch.counts().astype(np.int64)))
            TD.gather_counts(dist, counts_dev)
        return batch.info()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = dense_ms = cand_ms = 0.0
    binfo = None
    for _ in range(args.steps):
        binfo = step()
        kernel_ms += binfo["last_run_ms"]
        dense_ms += binfo["dense_ms"]
        cand_ms += binfo["cand_ms"]
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        m = torch.tensor([float(binfo["matches"]), float(binfo["algorithmic_bytes"])], dtype=torch.float64, device="cuda")
        dist.all_reduce(m, op=dist.ReduceOp.SUM)
        matches_all, alg_all = float(m[0].item()), float(m[1].item())
    else:
        matches_all, alg_all = float(binfo["matches"]), float(binfo["algorithmic_bytes"])

    if rank == 0:
        steps = max(1, args.steps)
        ms_per_step = elapsed * 1e3 / steps
        qps = args.queries * world * steps / elapsed
        # the step launches two matching kernels back to back on the engine stream; the dominant one carries the roofline
        kms = {"k_and_dense": dense_ms / steps, "k_and": cand_ms / steps}
        kalg = {"k_and_dense": float(binfo["dense_algorithmic_bytes"]), "k_and": float(binfo["cand_algorithmic_bytes"])}
        dom = max(kms, key=kms.get)
        achieved = kalg[dom] / (kms[dom] * 1e-3) / 1e9 if kms[dom] > 0 else 0.0
        k_ms = kernel_ms / steps
        alg = float(binfo["algorithmic_bytes"])
        traffic = pmc_traffic(args, world) if progs is None else None
        if progs is not None:
            # extra workloads launch more kernels (k_phrase, k_score, k_topk_merge): whole-step figures only
            dom = "whole step"
            kms = {dom: k_ms}
            kalg = {dom: alg}
            achieved = alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            "metric": "queries/sec",
            "value": qps,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": (wl_desc + f", Zipf(1.0) {args.docs} docs / {args.vocab} terms") if wl_desc else ("cfg2: batched 2-term AND, google_codec, DocumentsOnly, Zipf(1.0) 10M docs / 1M terms" if args.docs == 10_000_000 else f"2-term AND, google_codec, DocumentsOnly, {args.docs} docs / {args.vocab} terms"),
                "docs": args.docs,
                "vocab": args.vocab,
                "queries_per_gpu_per_step": args.queries,
                "index_bytes": int(info["index_bytes"]),
                "postings": int(info["postings"]),
                "parallelism": f"query-sharded x{world}, index replicated",
            },
            "matched_docids_per_sec": matches_all * steps / elapsed,
            "matches_per_step": matches_all,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": (traffic or {}).get(dom),
                "kernel": dom,
                "kernel_ms": kms[dom],
                "algorithmic_bytes_per_launch": kalg[dom],
                "queries_per_launch": int(binfo["dense_queries"] if dom == "k_and_dense" else binfo["cand_queries"] if dom == "k_and" else args.queries),
                "other_kernels": {k: {"kernel_ms": kms[k], "algorithmic_bytes_per_launch": kalg[k], "achieved": (kalg[k] / (kms[k] * 1e-3) / 1e9 if kms[k] > 0 else 0.0), "traffic": (traffic or {}).get(k)} for k in kms if k != dom},
                "whole_step": {"kernel_ms": k_ms, "algorithmic_bytes": alg, "achieved": alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0, "frac": (alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_ms > 0 else 0.0},
            },
            "segment_build_s": build_s,
            "index_upload_s": upload_s,
        }
        if args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(seg, qs, args.cpu_seconds) if progs is None else cpu_baseline_programs(seg, progs, wflags, args.cpu_seconds)
            # the sample the oracle just ran is also a full-size parity check: same queries, same segment, match totals must agree
            n_s = out["cpu_baseline"].pop("_n", 0)
            m_s = out["cpu_baseline"].pop("_matches", None)
            if n_s and m_s is not None:
                gpu_m = int(batch.counts()[:n_s].astype(np.uint64).sum())
                out["parity_check"] = {"queries": n_s, "cpu_oracle_matches": int(m_s), "gpu_matches": gpu_m, "equal": bool(gpu_m == int(m_s))}
        print(json.dumps(out), flush=True)

    batch.close()
    ix.close()
    dev.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(args, world):
    """HBM bytes per launch of each matching kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950
    correction + WRITE_SIZE; profiles/pmc_latest.json, collected with this exact workload) — PMC counters cannot be read
    from inside the timed run, so the figures are reported only when the configuration matches."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            p = json.load(f)["bench_cfg2"]
        if (p["docs"], p["vocab"], p["queries"]) == (args.docs, args.vocab, args.queries):
            return {k: v["traffic_bytes_per_launch"] for k, v in p["kernels"].items()}
    except Exception:
        pass
    return None


def cpu_baseline_programs(seg, progs, flags, budget_s):
    """Extra workloads: the CPU oracle, one thread, on the first programs of rank 0's batch until the budget is spent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    if seg.codec == 2:
        ora = O.Index.generate(seg.D, seg.V, seg.slots, seg.seed, codec="lucene")
    else:
        ora = O.Index.wrap(seg.index, seg.terms, seg.docs_cnt, seg.sum_terms_docs, seg.sum_term_hits)
    n = matches = 0
    t0 = time.perf_counter()
    for p in progs:
        matches += ora.exec_count(p, flags)
        n += 1
        if time.perf_counter() - t0 > budget_s and n >= 16:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": f"first {n} programs of rank 0's batch ({matches} matches) in {dt:.1f}s, oracle single thread", "host_cpus": os.cpu_count(),
            "_n": n, "_matches": matches}


def cpu_baseline(seg, qs, budget_s):
    """The CPU oracle (plain-C restatement of the reference's iterator path: Google::Decoder next/advance ->
    Conjuction leapfrog -> GenericDocsSetSpan), one thread, on the first queries of the same batch until the time
    budget is spent.  A reported baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O

    ora = O.Index.wrap(seg.index, seg.terms, seg.docs_cnt, seg.sum_terms_docs, seg.sum_term_hits)
    n = 0
    matches = 0
    t0 = time.perf_counter()
    for a, b in qs.tolist():
        docs, _ = ora.exec(np.array([O.tok(O.OP_TERM, a), O.tok(O.OP_TERM, b), O.tok(O.OP_AND, 2)], dtype=np.uint32), O.FLAG_DOCUMENTS_ONLY)
        matches += len(docs)
        n += 1
        if time.perf_counter() - t0 > budget_s and n >= 32:
            break
    dt = time.perf_counter() - t0
    res = {
        "value": n / dt,
        "unit": "queries/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {n} queries of rank 0's batch ({matches} matches) in {dt:.1f}s, oracle/trinity_oracle.c single thread",
        "matched_docids_per_sec": matches / dt,
        "host_cpus": os.cpu_count(),
        "_n": n,
        "_matches": matches,
    }
    # also one query per thread on all host cores (the reference's exec_query is re-entrant per thread,
    # exec.cpp:12).  Same oracle, same queries, drawn from a shared cursor by C threads.
    try:
        ncores = len(os.sched_getaffinity(0))
        progs = np.array([[O.tok(O.OP_TERM, a), O.tok(O.OP_TERM, b), O.tok(O.OP_AND, 2)] for a, b in qs.tolist()], dtype=np.uint32)
        # heaviest queries first, so the figure is only meaningful over the WHOLE batch: the budget is a safety net, not a cut
        done, m, dt2 = ora.exec_batch_mt(progs, O.FLAG_DOCUMENTS_ONLY, ncores, max(30.0, budget_s * 3))
        res["all_cores"] = {
            "value": done / dt2,
            "unit": "queries/s",
            "cores": ncores,
            "sample": f"{done} queries of the same batch ({m} matches) in {dt2:.1f}s, one query per thread (pthreads, oracle to_exec_batch_mt)",
            "matched_docids_per_sec": m / dt2,
        }
    except Exception as e:  # the single-thread figure stands on its own
        res["all_cores"] = {"error": str(e)}
    return res


if __name__ == "__main__":
    main()
