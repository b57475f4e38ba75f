"""One rank of the multi-rank GPU communicator test (started by hippyflow_amd.launch.spawn_ranks).  On a one-GPU box
the ranks share GPU 0 and the communicator picks the p2p transport (HIP IPC staging buffers, reduction on the
device); with one GPU per rank it picks RCCL.  Checks, per rank:
  * block all-reduce (sum / avg) and broadcast against blocks regenerated locally (the generator is counter based);
  * the sample-sharded fused double pass (all-reduce enqueued by the C solve) == the one-rank solve over all samples."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    outdir = sys.argv[1]
    import hippyflow_amd as hf
    from hippyflow_amd import workloads
    from hippyflow_amd.randomized import _ParRandom
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    coll = hf.NativeCollective.from_env()
    res = {"size": coll.size(), "rank": coll.rank(), "transport": coll.transport, "p2p_sync": coll.describe()["p2p_sync"]}
    ctx = hf.Context.default()

    N, k = 5003, 7

    def block_of(r):
        mv = hf.MultiVector(N, k, ctx=ctx)
        _ParRandom(100 + r).normal(1.0, mv)
        return mv

    mine = block_of(rank)
    parts = [block_of(r).to_dense() for r in range(world)]
    expect = parts[0].copy()
    for p in parts[1:]:
        expect = expect + p                                  # rank order, as the transports sum
    y = hf.MultiVector(mine)
    coll.allReduce(y, "sum")
    res["sum_err"] = float(np.abs(y.to_dense() - expect).max())
    y = hf.MultiVector(mine)
    coll.allReduce(y, "AVG")
    res["avg_err"] = float(np.abs(y.to_dense() - expect / world).max() / np.abs(expect).max())
    res["avg_block"] = y.to_dense()[:64].copy()              # compared across ranks: identical bits
    y = hf.MultiVector(mine)
    coll.bcast(y, root=world - 1)
    res["bcast_err"] = float(np.abs(y.to_dense() - parts[world - 1]).max())
    v = y[2]
    coll.allReduce(v, "sum")                                  # a Vector (one column view of a block)
    res["vector_err"] = float(np.abs(v.get_local() - world * parts[world - 1][:, 2]).max())
    big = hf.MultiVector(20011, 33, ctx=ctx)                  # grows the staging buffers
    _ParRandom(7).normal(1.0, big)
    ref = big.to_dense()
    coll.allReduce(big, "avg")
    res["big_err"] = float(np.abs(big.to_dense() - ref).max() / np.abs(ref).max())
    res["host_sum"] = coll.allReduce(float(rank + 1), "sum")

    # sample-sharded fused solve vs the one-rank solve over all samples
    Nn, ns_total, q, r, p = 3001, 8, 6, 4, 3
    ns_local = ns_total // world
    wl = workloads.as_workload(Nn, ns_local, q=q, latent=q, rate=0.3, seed=4, first_sample=rank * ns_local, ns_total=ns_total, ctx=ctx)
    hf.parRandom.reseed(11)
    Omega = hf.MultiVector(Nn, r + p, ctx=ctx)
    hf.parRandom.normal(1.0, Omega)
    for name, kw in (("gram", {}), ("literal", {"literal_T": True})):
        d, U = hf.doublePass(hf.CollectiveOperator(wl.operator, coll, mpi_op="avg"), Omega, r, s=1, **kw)
        res["d_" + name] = d
        res["U_" + name] = U.to_dense()[:50].copy()
    d_cols, _ = hf.doublePass(hf.MatrixMultCollectiveOperator(wl.operator, coll, mpi_op="avg"), Omega, r, s=1, fused=False)
    res["d_unfused"] = d_cols
    if rank == 0:
        wl_all = workloads.as_workload(Nn, ns_total, q=q, latent=q, rate=0.3, seed=4, first_sample=0, ns_total=ns_total, ctx=ctx)
        d_all, U_all = hf.doublePass(wl_all.operator, Omega, r, s=1)
        res["d_all"], res["U_all"] = d_all, U_all.to_dense()[:50].copy()
    coll.barrier()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **res)
    coll.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
