"""Multi-GPU plumbing: one process per GPU, index replicated, the query stream sharded, results gathered.

Queries are independent units (reference: exec.h:57-62, index_source.h:210-211 — per-source results only need a
merge), so there is NO data-path collective.  The only exchanges are fixed-shape result blocks at the end of a
batch: per-query match counts (DocumentsOnly) and `[Q/G][K]` top-K docID/score blocks (AccumulatedScoreScheme),
all-gathered with torch.distributed — backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import numpy as np


def shard_rows(rows, rank, world, per_rank=None):
    """Interleaved shard of a query table: rank r takes rows r, r+world, … (every rank sees the same cost mix)."""
    s = rows[rank::world]
    return s if per_rank is None else s[:per_rank]


def unshard_index(nq_total, rank, world):
    """Global query indices owned by `rank` under shard_rows."""
    return np.arange(rank, nq_total, world)


def gather_counts(dist, counts_t):
    """all_gather of per-query match counts; returns a list of per-rank tensors (same device as counts_t)."""
    out = [counts_t.new_zeros(counts_t.shape) for _ in range(dist.get_world_size())]
    dist.all_gather(out, counts_t)
    return out


def gather_topk(dist, docs_t, scores_t, counts_t):
    """all_gather of the fixed-shape top-K result blocks ([Q/G][K] u32-as-int32 docIDs, f32 scores, [Q/G] counts)."""
    w = dist.get_world_size()
    d = [docs_t.new_zeros(docs_t.shape) for _ in range(w)]
    s = [scores_t.new_zeros(scores_t.shape) for _ in range(w)]
    c = [counts_t.new_zeros(counts_t.shape) for _ in range(w)]
    dist.all_gather(d, docs_t)
    dist.all_gather(s, scores_t)
    dist.all_gather(c, counts_t)
    return d, s, c


def interleave(per_rank_arrays):
    """Undo shard_rows: per-rank arrays (equal leading length) -> global query order."""
    w = len(per_rank_arrays)
    n = sum(a.shape[0] for a in per_rank_arrays)
    out = np.zeros((n,) + tuple(per_rank_arrays[0].shape[1:]), dtype=per_rank_arrays[0].dtype)
    for r, a in enumerate(per_rank_arrays):
        out[r::w] = a
    return out
