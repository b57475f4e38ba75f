"""Projection-error tests of a computed basis: the immediate consumer of (d, U).

Reference: KLEProjector.test_errors (modeling/KLEProjector.py:202-282), the input half of
ActiveSubspaceProjector.test_errors (modeling/activeSubspaceProjector.py:1093-1124) and
PODProjector.test_output_errors (modeling/PODProjector.py:439-476): for each requested rank r, project every
test sample x onto span(U_r) -- x_r = U_r U_r^T B x (B = M or prior precision for B-orthonormal bases, B = I
otherwise) -- and report mean and standard deviation of ||x - x_r||_2 / ||x||_2, averaged over ranks of the
collective.  On the device all samples are handled as one block: C = U^T (B X) is one reduction GEMM, each rank's
projection one expansion GEMM on the leading r vectors.
"""
import numpy as np

from .collectives import NullCollective
from .multivector import MatMvMult, MultiVector, MvDSmatMult
from .operators import as_device_operator


def projection_error_test(U, samples, ranks=(None,), B=None, d=None, cut_off=1e-12, collective=None):
    """U: MultiVector of basis vectors (ordered by decreasing eigenvalue); samples: MultiVector (or (n, N) array)
    of test vectors.  Returns (ranks_used, global_avg_rel_errors, global_std_rel_errors)."""
    collective = collective if collective is not None else NullCollective()
    if not isinstance(samples, MultiVector):
        samples = MultiVector.from_vectors(np.asarray(samples, dtype=np.float64), ctx=U.ctx)
    ranks = sorted(U.nvec() if r is None else int(r) for r in ranks)
    if d is not None:     # truncate for numerical stability, as the reference does (KLEProjector.py:227-229)
        numericalrank = int(np.where(np.asarray(d) > cut_off)[0][-1]) + 1
        ranks = [r for r in ranks if r <= numericalrank]
    ranks = [r for r in ranks if 0 < r <= U.nvec()]
    N, ns = samples.size(), samples.nvec()
    if B is not None:
        BX = MultiVector(N, ns, ctx=U.ctx)
        MatMvMult(as_device_operator(B, N, U.ctx), samples, BX)
    else:
        BX = samples
    C = U.dot_mv(BX)                      # (k, ns): coefficients of every sample on every basis vector
    denom = samples.norm()
    avg, std = np.ones(len(ranks)), np.zeros(len(ranks))
    E = MultiVector(N, ns, ctx=U.ctx)
    for idx, r in enumerate(ranks):
        MvDSmatMult(U.view(0, r), np.ascontiguousarray(C[:r]), E)        # projections
        E.scale(-1.0)
        E.axpy(1.0, samples)                                              # x - U_r U_r^T B x
        rel = E.norm() / denom
        avg[idx] = collective.allReduce(float(np.mean(rel)), 'avg')
        std[idx] = np.sqrt(collective.allReduce(float(np.std(rel) ** 2), 'avg'))
    return ranks, avg, std


def input_output_error_test(observable, prior, U_out, V_in, rank_pairs, n_samples, Cinv=None, noise=None, collective=None,
                            control_distribution=None):
    """PODProjector.input_output_error_test (modeling/PODProjector.py:541-655): how well the reduced map
    m -> U_r U_r^T q(V_s V_s^T [C^-1] m) reproduces q(m), for pairs (s, r) of input / output ranks, over ``n_samples`` prior
    draws.  The PDE solves stay the host's (``observable.solveFwd`` / ``evalu``, one per sample and one more per sample and
    rank pair, exactly the reference's count); everything between them is done on the whole batch on the device: the
    coefficients V^T (C^-1 M) of every sample on every input vector once, each pair's projected parameters as one expansion,
    each pair's output projection and error norms as two more.  Returns (global_avg_rel_errors, global_std_rel_errors)."""
    from . import hostvec as H
    from .multivector import ingest_stream
    from .randomized import parRandom
    assert control_distribution is None, 'Not worked out yet for control problems'        # as upstream (:550)
    collective = collective if collective is not None else NullCollective()
    ctx = U_out.ctx
    for rank_in, rank_out in rank_pairs:
        assert rank_in <= V_in.nvec() and rank_out <= U_out.nvec()
    if noise is None:
        noise = H.new_host_vector(observable.mpi_comm() if hasattr(observable, "mpi_comm") else None)
        prior.init_vector(noise, "noise")
    u, m = observable.generate_vector(H.STATE), observable.generate_vector(H.PARAMETER)
    n_in, n_out = V_in.size(), U_out.size()

    def pairs():                                            # the sampling loop (:571-581): (parameter, observable) rows
        for _ in range(n_samples):
            parRandom.normal(1, noise)
            prior.sample(noise, m)
            observable.solveFwd(u, [u, m, None])
            yield np.concatenate([m.get_local(), observable.evalu(u).get_local()])

    both = np.stack(list(pairs()))
    params = MultiVector.from_vectors(both[:, :n_in], ctx=ctx)
    obs_host = both[:, n_in:]
    observables = MultiVector.from_vectors(obs_host, ctx=ctx)
    if Cinv is not None:                                    # PriorPreconditionedProjector: V V^T C^-1 m (:600-601)
        CP = MultiVector(n_in, n_samples, ctx=ctx)
        MatMvMult(as_device_operator(Cinv, n_in, ctx), params, CP)
    else:
        CP = params
    coeff = V_in.dot_mv(CP)                                 # (k_in, n_samples)
    denom = observables.norm()
    projected = MultiVector(n_in, n_samples, ctx=ctx)
    avg, std = [], []
    for rank_in, rank_out in rank_pairs:
        MvDSmatMult(V_in.view(0, rank_in), np.ascontiguousarray(coeff[:rank_in]), projected)
        rows = projected.to_vectors()

        def reduced_observables():
            for i in range(n_samples):
                m.set_local(rows[i])
                m.apply("")
                observable.solveFwd(u, [u, m, None])
                yield observable.evalu(u).get_local()

        Qr = ingest_stream(reduced_observables(), n_samples, 1, n_out, ctx=ctx)
        Ur = U_out.view(0, rank_out)
        E = MultiVector(n_out, n_samples, ctx=ctx)
        MvDSmatMult(Ur, np.ascontiguousarray(Ur.dot_mv(Qr)), E)          # U_r U_r^T q_r
        E.scale(-1.0)
        E.axpy(1.0, observables)                                          # q - U_r U_r^T q_r
        rel = E.norm() / denom
        avg.append(collective.allReduce(float(np.mean(rel)), 'avg'))
        std.append(float(np.sqrt(collective.allReduce(float(np.std(rel) ** 2), 'avg'))))
    return avg, std
