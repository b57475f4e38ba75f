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
