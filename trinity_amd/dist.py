"""Multi-GPU plumbing: one process per GPU, index replicated, the query stream sharded, results gathered.

Queries are independent units (reference: exec.h:57-62, index_source.h:210-211 — per-source results only need a
merge; exec_query_par, exec.h:132-176, collects one result vector per source), so there is NO data-path collective.
The only exchange is the result gather at the end of a step: fixed-shape blocks per rank — per-query match counts
u64[Q/G] and, for AccumulatedScoreScheme top-K batches, [Q/G][K] docIDs + scores and [Q/G] list lengths — sent with
torch.distributed all_gather: backend "nccl" (= RCCL over xGMI) straight from the engine's device-resident result
buffers (no host bounce), "gloo" on CPU tensors in the tests.  The same ResultGather drives both.
"""
import numpy as np


def shard_rows(rows, rank, world, per_rank=None):
    """Interleaved shard of a query table: rank r takes rows r, r+world, … (every rank sees the same cost mix)."""
    s = rows[rank::world]
    return s if per_rank is None else s[:per_rank]


def unshard_index(nq_total, rank, world):
    """Global query indices owned by `rank` under shard_rows."""
    return np.arange(rank, nq_total, world)


def interleave(per_rank_arrays):
    """Undo shard_rows: per-rank arrays (equal leading length) -> global query order."""
    w = len(per_rank_arrays)
    n = sum(a.shape[0] for a in per_rank_arrays)
    out = np.zeros((n,) + tuple(per_rank_arrays[0].shape[1:]), dtype=per_rank_arrays[0].dtype)
    for r, a in enumerate(per_rank_arrays):
        out[r::w] = a
    return out


class _DeviceBlock:
    """A typed view of raw device memory for torch (zero copy) through the CUDA array interface."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_blocks(batch, device):
    """The result blocks of an engine Batch as torch tensors over the engine's own device buffers (no copy): "counts" i64[nq];
    AccumulatedScore top-K batches add "docs" i32[nq][k] (docIDs, bit pattern of the u32), "scores" f32[nq][k], "topk_counts" i32[nq]."""
    import torch

    p = batch.device_results()
    nq, k = batch.nq, batch.topk
    out = {"counts": torch.as_tensor(_DeviceBlock(p["counts"], (nq,), "<i8"), device=device)}
    if "docs" in p:
        out["docs"] = torch.as_tensor(_DeviceBlock(p["docs"], (nq, k), "<i4"), device=device)
        out["scores"] = torch.as_tensor(_DeviceBlock(p["scores"], (nq, k), "<f4"), device=device)
        out["topk_counts"] = torch.as_tensor(_DeviceBlock(p["topk_counts"], (nq,), "<i4"), device=device)
    return out


class ResultGather:
    """The per-step result exchange: every rank contributes the same named fixed-shape blocks; step() gathers each with ONE
    all_gather_into_tensor into a preallocated [world, ...] receive tensor (no per-rank list, no copies on either side) and returns
    {name: tensor[world, ...]} — row r is rank r's block."""

    def __init__(self, dist, blocks):
        import torch

        self.dist, self.blocks = dist, dict(blocks)
        self.world = dist.get_world_size()
        # (the receive tensor in its concatenated form [world * n, ...] — the one layout every backend's all_gather_into_tensor takes —
        #  and, over the same memory, the [world, n, ...] view callers index by rank)
        self._flat = {name: torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for name, t in self.blocks.items()}
        self.recv = {name: f.view((self.world,) + tuple(self.blocks[name].shape)) for name, f in self._flat.items()}

    def rebind(self, blocks):
        """Point the gather at another batch's result blocks of the same shapes (a caller that compiles a batch per step gathers from a
        different batch every step; the receive buffers stay)."""
        assert {n: (tuple(t.shape), t.dtype) for n, t in blocks.items()} == {n: (tuple(t.shape), t.dtype) for n, t in self.blocks.items()}
        self.blocks = dict(blocks)

    def step(self):
        for name, t in self.blocks.items():
            self.dist.all_gather_into_tensor(self._flat[name], t)
        return self.recv

    def global_order(self, name):
        """The gathered block re-interleaved to global query order (numpy; test / verification helper)."""
        return interleave([t.cpu().numpy() for t in self.recv[name]])
