"""Worker for test_gather_results_two_ranks_one_gpu (launched by torch.distributed.run, two ranks, backend gloo, BOTH on cuda:0 —
the GPU box has one device).  Every rank runs its shard of a scored batch on the engine and calls the C-ABI gather,
tri_gather_results, through a communicator built with tri_comm_create_custom: the allgather of the device-resident result
blocks is carried by this test's own transport (device -> host, gloo all_gather, host -> device).  What the gather must get
right — which blocks, their sizes, the [nranks][...] layout of the receive buffers — is then checked at world_size 2 against
the unsharded batch.  (The RCCL transport of the same call needs one device per rank; it runs with one rank in
test_gather_results_over_rccl_one_rank.)"""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import trinity_amd as T  # noqa: E402
from trinity_amd import dist as TD  # noqa: E402
from trinity_amd import engine as E  # noqa: E402


class _Dev:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    D, V, NQ, K = 20000, 2000, 64, 10
    seg = T.Segment(D, V, 10, 42)
    dev = T.Device(0)
    ix = T.Index.from_segment(dev, seg)
    texts = [f"t{a} OR t{b} OR t{c}" if i % 3 == 0 else f"t{a} t{b}" if i % 3 == 1 else f"t{a} (t{b} OR t{c})" for i, (a, b, c) in enumerate(T.gen_queries(V, 77, NQ, 3).tolist())]
    progs = [O.parse_query(t) for t in texts]
    mine = progs[rank::world]
    b = T.Batch(ix, mine, T.FLAG_ACCUMULATED_SCORE, topk=K)
    b.run()
    L = E.hip_lib()
    calls = []

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
    def allgather(user, send, recv, nbytes, stream):
        torch.cuda.synchronize()  # (the engine's stream is not torch's: complete before the copy)
        s = torch.as_tensor(_Dev(send, nbytes), device="cuda:0").cpu()
        parts = [torch.empty_like(s) for _ in range(world)]
        dist.all_gather(parts, s)
        torch.as_tensor(_Dev(recv, nbytes * world), device="cuda:0").copy_(torch.cat(parts).cuda())
        torch.cuda.synchronize()
        calls.append(nbytes)
        return 0

    comm = C.c_void_p()
    E._check(L.tri_comm_create_custom(dev.h, rank, world, allgather, None, C.byref(comm)))
    nq = len(mine)
    recv = [torch.zeros(world * n, dtype=torch.uint8, device="cuda:0") for n in (nq * 8, nq * K * 4, nq * K * 4, nq * 4)]
    E._check(L.tri_gather_results(b.h, comm, *[C.c_void_p(t.data_ptr()) for t in recv]))
    b.sync()
    dev.sync()
    assert calls == [nq * 8, nq * K * 4, nq * K * 4, nq * 4], calls
    got = [t.cpu().numpy().view(dt).reshape((world, nq) + shp) for t, dt, shp in zip(recv, (np.uint64, np.uint32, np.float32, np.uint32), ((), (K,), (K,), ()))]
    # every rank holds every rank's blocks; re-interleaved they are the unsharded batch's results
    full = T.Batch(ix, progs, T.FLAG_ACCUMULATED_SCORE, topk=K)
    full.run()
    full.sync()
    d, s, c = full.topk_results()
    assert np.array_equal(TD.interleave(list(got[0])), full.counts())
    assert np.array_equal(TD.interleave(list(got[1])), d) and np.array_equal(TD.interleave(list(got[2])), s) and np.array_equal(TD.interleave(list(got[3])), c)
    L.tri_comm_destroy(comm)
    full.close()
    b.close()
    ix.close()
    dev.close()
    dist.barrier()
    if rank == 0:
        with open(out_path, "w") as f:
            f.write(f"ok world={world} queries={NQ}\n")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
