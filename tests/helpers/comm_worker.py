"""One rank of the native-communicator CPU test (started by hippyflow_amd.launch.spawn_ranks): a host-only
hfmi_comm (no GPU) exercising the id exchange, the shared-segment barrier and the host-payload collectives with the
payload rules of the reference (collective.py:74-159)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    outdir, mode = sys.argv[1], sys.argv[2]
    from hippyflow_amd.collectives import CollectiveOperator, MatrixMultCollectiveOperator, NativeCollective
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if mode == "missing_peer" and rank == world - 1:
        return 0                                     # this rank never joins: the others must time out, not hang
    if mode == "back_to_back":
        # communicators made one after the other through the SAME id file must never meet in a stale segment
        sums = []
        for it in range(6):
            c = NativeCollective.from_env(host_only=True)
            sums.append(c.allReduce(float(rank + 1 + it), "sum"))
            if it % 2 == 0:
                c.close()               # some are closed at once, some stay open while the next one is made
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), sums=np.array(sums), why=c.describe()["why"])
        return 0
    coll = NativeCollective.from_env(host_only=True)
    res = {"size": coll.size(), "rank": coll.rank(), "transport": coll.transport}
    if mode == "missing_peer":
        return 0
    res["sum_float"] = coll.allReduce(float(rank + 1), "sum")
    res["avg_float"] = coll.allReduce(float(rank + 1), "Avg")
    res["sum_int"] = coll.allReduce(int(rank + 1), "sum")
    res["sum_int_is_int"] = isinstance(res["sum_int"], int)
    res["avg_int"] = coll.allReduce(int(rank + 1), "avg")
    big = np.arange(20000, dtype=np.float64) * (rank + 1)          # more than one 8192-double chunk
    out = coll.allReduce(big, "avg")
    res["big_avg_err"] = float(np.abs(out - np.arange(20000) * (world + 1) / 2.0).max())
    res["big_inplace"] = out is big
    strided = np.zeros((6, 4))[:, ::2]
    strided[...] = rank + 1.0
    coll.allReduce(strided, "sum")
    res["strided"] = strided.copy()
    b = np.full(9000, float(rank))
    coll.bcast(b, root=world - 1)
    res["bcast_ok"] = bool((b == world - 1).all())
    ints = np.array([rank, 7 * rank + 3], dtype=np.uint64)
    coll.bcast(ints, root=0)
    res["bcast_ints"] = ints.copy()
    res["bcast_scalar"] = coll.bcast(float(rank) + 0.5, root=0)
    res["bcast_int_scalar"] = coll.bcast(int(rank) + 5, root=world - 1)
    res["max"] = coll.allReduceMax(10.0 * rank)
    for bad in ("max", None):
        try:
            coll.allReduce(np.zeros(2), bad)
            res["bad_op_%s" % bad] = False
        except NotImplementedError:
            res["bad_op_%s" % bad] = True
    try:
        coll.allReduce("a string", "sum")
        res["bad_type"] = False
    except NotImplementedError:
        res["bad_type"] = True
    coll.barrier()

    # the averaging identity of the sample-parallel path with the oracle's operators (host arrays as "vectors")
    from hippyflow_amd import workloads
    from oracle import hippyflow_restated as hf_o
    from oracle import hippylib_restated as hp_o
    ns_total, q, c, N, k, r = 12, 5, 5, 80, 6, 3
    ns_local = ns_total // world
    rng = np.random.default_rng(0)
    P, _ = np.linalg.qr(rng.standard_normal((N, c)))
    s = np.exp(-0.3 * np.arange(c))
    J_local = np.stack([(workloads.sample_factor(4, rank * ns_local + i, q, c) * s) @ P.T for i in range(ns_local)])
    Omega = np.asfortranarray(np.random.default_rng(1).standard_normal((N, k)))

    class NumpyBlockOp:
        def __init__(self, J):
            self.op = hf_o.MeanJTJOperator(J)

        def mult(self, x, y):
            self.op.mult(x, y)

        def matMvMult(self, X, Y):
            Y[...] = 0.0
            self.op.matMvMult(X, Y)

    class FBlock(np.ndarray):
        pass

    d_par, _ = hp_o.double_pass(_Contig(MatrixMultCollectiveOperator(NumpyBlockOp(J_local), coll, mpi_op="avg")), Omega, r)
    res["d_par"] = d_par
    if rank == 0:
        J_all = np.stack([(workloads.sample_factor(4, i, q, c) * s) @ P.T for i in range(ns_total)])
        res["d_ser"], _ = hp_o.double_pass(hf_o.MeanJTJOperator(J_all), Omega, r)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **res)
    coll.close()
    return 0


class _Contig:
    """The oracle hands Fortran-ordered blocks to matMvMult; the collective reduces any layout in place."""

    def __init__(self, op):
        self.op = op

    def matMvMult(self, X, Y):
        self.op.matMvMult(X, Y)


if __name__ == "__main__":
    sys.exit(main())
