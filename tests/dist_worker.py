"""Worker for tests/test_dist_gloo.py (launched by torch.distributed.run, backend gloo, CPU only).
Each rank shards the query table exactly as bench.py does, produces its shard's results — here with the CPU
oracle standing in for the engine, because there is no GPU in this test — and the ranks exchange the fixed-shape
result blocks with trinity_amd.dist.  Rank 0 checks the gathered, re-interleaved results against the unsharded
answers and writes a marker file."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
from trinity_amd import dist as TD  # noqa: E402


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    D, V, NQ, K = 20000, 2000, 64, 10
    ix = O.Index.generate(D, V, 10, 42)
    qall = O.gen_queries(V, 1337, NQ, 2)
    mine = TD.shard_rows(qall, rank, world)

    def answer(rows):
        counts = np.zeros(len(rows), dtype=np.int64)
        docs = np.zeros((len(rows), K), dtype=np.int32)
        scores = np.zeros((len(rows), K), dtype=np.float32)
        tc = np.zeros(len(rows), dtype=np.int32)
        for i, (a, b) in enumerate(rows.tolist()):
            d, s = ix.exec(np.array([O.tok(O.OP_TERM, a), O.tok(O.OP_TERM, b), O.tok(O.OP_AND, 2)], dtype=np.uint32), O.FLAG_ACCUM_SCORE)
            counts[i] = len(d)
            td, ts = ix.topk(d, s, K)
            docs[i, : len(td)] = td.astype(np.int32)
            scores[i, : len(td)] = ts
            tc[i] = len(td)
        return counts, docs, scores, tc

    counts, docs, scores, tc = answer(mine)
    # the same ResultGather bench.py drives over the engine's device buffers (there: backend nccl = RCCL, tensors from device_blocks)
    g = TD.ResultGather(dist, {"counts": torch.from_numpy(counts), "docs": torch.from_numpy(docs), "scores": torch.from_numpy(scores), "topk_counts": torch.from_numpy(tc)})
    for _ in range(2):  # a step can be repeated: the receive buffers are reused
        g.step()
    if rank == 0:
        all_counts, all_docs, all_scores, all_tc = (g.global_order(n) for n in ("counts", "docs", "scores", "topk_counts"))
        want = answer(qall)
        assert np.array_equal(all_counts, want[0])
        assert np.array_equal(all_docs, want[1])
        assert np.array_equal(all_scores, want[2])
        assert np.array_equal(all_tc, want[3])
        for r in range(world):
            assert np.array_equal(TD.unshard_index(NQ, r, world), np.arange(NQ)[r::world])
        with open(out_path, "w") as f:
            f.write(f"ok world={world} queries={NQ} matches={int(all_counts.sum())}\n")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
