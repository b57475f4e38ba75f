"""Second worker of the multi-rank GPU communicator tests (ranks share GPU 0: p2p transport).  Modes (argv[2]):

  late    rank 1 arrives at a block all-reduce seconds after the time-out: rank 0 must get HFMI_ERR_COMM at its next host
          synchronisation instead of an unreduced block, and the late rank must fail too (it meets rank 0's poison flag)
          rather than walk away with a half-reduced one.
  panels  a Gram-form fused solve with at least two rounds of row tiles, so that the rank reduction runs panel by panel on
          the auxiliary stream across REAL peers; the result with 4 panels must equal the result with one all-reduce bit for
          bit, on every rank.
  grow    block all-reduces of GROWING sizes (1, 3, 3.9, 9 MB per rank) while HFMI_P2P_INJECT_EXPORT_FAIL makes one rank retry the export
          of its first staging buffer (its allocation is then 2 MB larger than its peers'): every rank must still take the same decision
          about when the buffers grow -- each reduction complete and correct on every rank, no time-out.
  soak    200 block all-reduces in a row (alternating sum / avg, two block sizes) of the same rank-specific inputs: every
          repetition must give the bits of the first one, on every rank (fixed summation order of the p2p reduction)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    outdir, mode = sys.argv[1], sys.argv[2]
    import hippyflow_amd as hf
    from hippyflow_amd import _lib as L
    from hippyflow_amd import workloads
    from hippyflow_amd.randomized import _ParRandom
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    coll = hf.NativeCollective.from_env()
    ctx = hf.Context.default()
    res = {"rank": rank, "transport": coll.transport, "p2p_sync": coll.describe()["p2p_sync"]}
    if mode == "late":
        X = hf.MultiVector(4001, 5, ctx=ctx)
        _ParRandom(50 + rank).normal(1.0, X)
        coll.allReduce(X, "sum")                     # a first collective that everybody reaches: staging buffers exist
        ctx.synchronize()
        coll.barrier()
        os.environ["HFMI_COMM_TIMEOUT_S"] = os.environ.get("HFMI_TEST_TIMEOUT_S", "1")    # read by libhfmi at every wait: short from here on
        if rank == 1:
            time.sleep(float(os.environ.get("HFMI_TEST_LATE_S", "4")))
        try:
            coll.allReduce(X, "sum")
            X.to_dense()                             # host synchronisation: the error surfaces here at the latest
            res["outcome"] = "returned a block"
        except hf.HfmiError as exc:
            res["outcome"] = "HfmiError"
            res["message"] = str(exc)
        with open(os.path.join(outdir, "late_rank%d.json" % rank), "w") as f:
            json.dump(res, f)
        os._exit(0)                                  # the communicator is unusable: no barrier, no orderly close
    if mode == "grow":
        sums = []
        for nrows in (13001, 39001, 51001, 118001):           # x 10 columns x 8 bytes: 1.0, 3.1, 4.1, 9.4 MB
            X = hf.MultiVector(nrows, 10, ctx=ctx)
            _ParRandom(70).normal(1.0, X)                     # the same block on every rank ...
            ref = X.to_dense() * float(sum(r + 1 for r in range(world)))
            X.scale(float(rank + 1))                          # ... scaled by rank + 1: the sum is known
            coll.allReduce(X, "sum")
            got = X.to_dense()
            sums.append(float(np.abs(got - ref).max() / np.abs(ref).max()))
        res["max_rel_err"] = sums
        res["describe"] = coll.describe()
        with open(os.path.join(outdir, "grow_rank%d.json" % rank), "w") as f:
            json.dump(res, f)
        coll.barrier()
        coll.close()
        return
    if mode == "panels":
        # per rank m = 16 x 100 rows of J: too many for the LDS-resident product (the hook lives in the tiled one), and N gives it
        # at least two rounds of row tiles whatever tile height the plan picks
        N, ns_total, q, k, r = 400000, 16 * world, 100, 12, 8
        ns_local = ns_total // world
        wl = workloads.as_workload(N, ns_local, q=q, latent=20, rate=0.2, seed=4, first_sample=rank * ns_local, ns_total=ns_total, ctx=ctx)
        Omega = hf.MultiVector(N, k, ctx=ctx)
        _ParRandom(11).normal(1.0, Omega)
        A = hf.CollectiveOperator(wl.operator, coll, mpi_op="avg")
        out = {}
        for panels in (0, 4):
            L.call("hfmi_tuning_set", b"comm_panels", panels)
            ctx.profile_begin()
            d, U = hf.doublePass(A, Omega, r, s=1)
            ctx.profile_end()
            out[panels] = (d, U.to_dense()[:4000].copy(), ctx.profile_phases()["allreduce_overlapped"])
        np.savez(os.path.join(outdir, "panels_rank%d.npz" % rank), d0=out[0][0], U0=out[0][1], d4=out[4][0], U4=out[4][1],
                 overlapped0=out[0][2], overlapped4=out[4][2], transport=coll.transport, p2p_sync=res["p2p_sync"])
        coll.barrier()
        coll.close()
        return 0
    if mode == "soak":
        import hashlib
        reps = int(os.environ.get("HFMI_TEST_SOAK_REPS", "200"))
        shapes = [(100003, 7), (33333, 74)]
        srcs = []
        for i, (N, k) in enumerate(shapes):
            X = hf.MultiVector(N, k, ctx=ctx)
            _ParRandom(70 + 10 * i + rank).normal(1.0, X)
            srcs.append(X)
        first, ref, same, bad = {}, {}, True, []
        for rep in range(reps):
            i, op = rep & 1, ("sum", "avg")[(rep >> 1) & 1]
            Y = hf.MultiVector(srcs[i])                     # a fresh copy of this rank's input
            coll.allReduce(Y, op)
            got = np.ascontiguousarray(Y.to_dense())
            digest = hashlib.sha256(got.tobytes()).hexdigest()
            if (i, op) in first:
                if first[(i, op)] != digest:
                    same = False
                    diff = got != ref[(i, op)]
                    rows = np.nonzero(diff.any(axis=1))[0]
                    bad.append({"rep": rep, "shape": i, "op": op, "entries": int(diff.sum()), "first_row": int(rows[0]), "last_row": int(rows[-1]),
                                "max_abs": float(np.abs(got - ref[(i, op)]).max())})
            else:
                first[(i, op)], ref[(i, op)] = digest, got
        desc = coll.describe()
        with open(os.path.join(outdir, "soak_rank%d.json" % rank), "w") as f:
            json.dump(dict(res, reps=reps, same=bool(same), mismatches=bad[:20], digests={"%d%s" % key: v for key, v in first.items()},
                           probe_rounds=desc.get("p2p_probe_rounds"), probe_retries_total=desc.get("p2p_regenerations_total")), f)
        coll.barrier()
        coll.close()
        return 0
    raise SystemExit("unknown mode " + mode)


if __name__ == "__main__":
    sys.exit(main())
