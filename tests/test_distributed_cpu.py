"""Multi-process CPU tests (gloo, world_size 2) of the sample-parallel path: the collective classes and the
averaging identity the 'avg' collective relies on (activeSubspaceProjector.py:429-430,509-511):

    P ranks, each averaging its own equal-sized shard of samples, all-reduced with 'avg'
        ==  one rank averaging all samples.

The local operators here are the CPU oracle's (the device kernels need a GPU); the classes under test --
TorchCollective, CollectiveOperator, MatrixMultCollectiveOperator and bench.py's shard arithmetic -- are the
product's own."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hippyflow_amd.collectives import CollectiveOperator, MatrixMultCollectiveOperator, TorchCollective
    from hippyflow_amd import workloads
    from oracle import hippyflow_restated as hf_o
    from oracle import hippylib_restated as hp_o
    res = {}
    coll = TorchCollective()
    res["size"], res["rank"] = coll.size(), coll.rank()
    # scalar / array semantics of collective.py:74-117
    res["sum_float"] = coll.allReduce(float(rank + 1), "sum")
    res["avg_float"] = coll.allReduce(float(rank + 1), "Avg")
    res["sum_int"] = coll.allReduce(int(rank + 1), "sum")
    arr = np.arange(6, dtype=np.float64) * (rank + 1)
    out = coll.allReduce(arr, "avg")
    res["arr_avg"], res["arr_inplace"] = out.copy(), out is arr
    b = np.full(4, float(rank))
    coll.bcast(b, root=1)
    res["bcast"] = b.copy()
    res["bcast_scalar"] = coll.bcast(float(rank) + 0.5, root=0)
    for bad in ("max",):
        try:
            coll.allReduce(np.zeros(2), bad)
            res["bad_op"] = False
        except NotImplementedError:
            res["bad_op"] = True
    try:
        coll.allReduce("a string", "sum")
        res["bad_type"] = False
    except NotImplementedError:
        res["bad_type"] = True
    # a collective keys the process-wide generator's PRIVATE streams by its rank; SHARED streams (probe blocks) stay common
    from hippyflow_amd.randomized import parRandom
    res["random_rank"], res["key_private"], res["key_shared"] = parRandom.rank, parRandom.key(False), parRandom.key(True)

    # averaging identity with the workload's own sample keying (global sample index -> factor)
    ns_total, q, c, N, k, r = 8, 6, 6, 90, 7, 4
    ns_local = ns_total // world
    rng = np.random.default_rng(0)
    P, _ = np.linalg.qr(rng.standard_normal((N, c)))
    s = np.exp(-0.3 * np.arange(c))
    J_local = np.stack([(workloads.sample_factor(4, rank * ns_local + i, q, c) * s) @ P.T for i in range(ns_local)])
    Omega = np.asfortranarray(np.random.default_rng(1).standard_normal((N, k)))

    class NumpyBlockOp:                      # oracle operator behind the reference protocol, numpy "vectors"
        def __init__(self, J):
            self.op = hf_o.MeanJTJOperator(J)

        def mult(self, x, y):
            self.op.mult(x, y)

        def matMvMult(self, X, Y):
            Y[...] = 0.0
            self.op.matMvMult(X, Y)

    d_par, U_par = hp_o.double_pass(MatrixMultCollectiveOperator(NumpyBlockOp(J_local), coll, mpi_op="avg"), Omega, r)
    d_par2, _ = hp_o.double_pass(_ColumnOnly(CollectiveOperator(NumpyBlockOp(J_local), coll, mpi_op="avg")), Omega, r)
    res["d_par"], res["d_par_columns"], res["U_par"] = d_par, d_par2, U_par
    if rank == 0:
        J_all = np.stack([(workloads.sample_factor(4, i, q, c) * s) @ P.T for i in range(ns_total)])
        d_ser, U_ser = hp_o.double_pass(hf_o.MeanJTJOperator(J_all), Omega, r)
        res["d_ser"], res["U_ser"] = d_ser, U_ser
        H = workloads.as_reduced_matrix(4, ns_total, q, c, 0.3)
        res["H_err"] = np.abs(P @ H @ P.T - np.einsum("iod,ioe->de", J_all, J_all) / ns_total).max()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **res)
    dist.barrier()
    dist.destroy_process_group()


class _ColumnOnly:
    """Hide matMvMult so that hp.MatMvMult falls back to the per-column loop of the reference."""

    def __init__(self, op):
        self.op = op

    def mult(self, x, y):
        self.op.mult(x, y)


def test_two_rank_collective_and_averaging_identity(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "rank1.npz"))
    for r, rank in ((r0, 0), (r1, 1)):
        assert int(r["size"]) == 2 and int(r["rank"]) == rank
        assert float(r["sum_float"]) == 3.0 and float(r["avg_float"]) == 1.5 and int(r["sum_int"]) == 3
        np.testing.assert_allclose(r["arr_avg"], np.arange(6) * 1.5)
        assert bool(r["arr_inplace"]) and bool(r["bad_op"]) and bool(r["bad_type"])
        np.testing.assert_array_equal(r["bcast"], np.full(4, 1.0))
        assert float(r["bcast_scalar"]) == 0.5
        assert int(r["random_rank"]) == rank and int(r["key_private"]) >> 32 == rank + 1 and int(r["key_shared"]) >> 32 == 0
    # both ranks hold the same result, and it equals the single-rank solve over all samples
    np.testing.assert_array_equal(r0["d_par"], r1["d_par"])
    np.testing.assert_allclose(r0["d_par"], r0["d_ser"], rtol=1e-12)
    np.testing.assert_allclose(r0["d_par_columns"], r0["d_ser"], rtol=1e-12)       # per-column all-reduce route
    np.testing.assert_allclose(np.abs(np.sum(r0["U_par"] * r0["U_ser"], axis=0)), 1.0, atol=1e-10)
    assert float(r0["H_err"]) < 1e-12                                               # bench.py's factored oracle operator


def test_bench_shard_arithmetic():
    """bench.py gives rank r the global samples [r*512/P, (r+1)*512/P): a partition for every P the driver uses."""
    for world in (1, 2, 4, 8):
        ns_local = 512 // world
        owned = sorted(i for r in range(world) for i in range(r * ns_local, (r + 1) * ns_local))
        assert owned == list(range(512))
