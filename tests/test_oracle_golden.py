"""Pin the oracle (oracle/hippyflow_restated.py) against vectors produced by the
reference's own numpy code (tests/golden/make_goldens.py), and against the
tolerances of the reference's tests (hippyflow/test/test_PODProjector.py:154-208)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import hippyflow_restated as hf_o
from oracle import hippylib_restated as hp_o


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _csr(g, N):
    return sp.csr_matrix((g["M_data"], g["M_indices"], g["M_indptr"]), shape=(N, N))


def _subspace_dist(A, B, M):
    """B-norm distance between spans of two M-orthonormal bases."""
    return hp_o.principal_angle(np.asfortranarray(A), np.asfortranarray(B), lambda X: M @ X)


@pytest.mark.parametrize("shifted", [True, False])
@pytest.mark.parametrize("method", ["hep", "ghep", "inverse_ghep"])
def test_pod_from_data_matches_reference(golden_dir, method, shifted):
    g = _load(golden_dir, "pod_from_data.npz")
    N, r = int(g["N"]), int(g["r"])
    M = _csr(g, N)
    tag = "%s_%d" % (method, int(shifted))
    d, phi, Mphi, shift = hf_o.pod_from_data(g["u_data"].copy(), M, r, shifted=shifted, method=method)
    np.testing.assert_allclose(shift, g["shift_" + tag], rtol=0, atol=1e-14)
    # leading eigenvalues to near round-off; the tail of an exp(-0.7 j) spectrum
    # only to the absolute accuracy eigh/eigsh deliver
    np.testing.assert_allclose(d, g["d_" + tag], rtol=1e-7, atol=1e-12 * abs(g["d_" + tag][0]))
    # eigenvectors: sign is arbitrary (eigsh start vector) -> compare subspaces
    # of the well-separated leading modes and |cosine| per mode
    lead = 6
    assert _subspace_dist(phi[:, :lead], g["phi_" + tag][:, :lead], M) < 1e-6
    cosines = np.abs(np.einsum("ij,ij->j", phi[:, :lead], M @ g["phi_" + tag][:, :lead]))
    np.testing.assert_allclose(cosines, 1.0, atol=1e-8)
    # the reference's own invariants (test_PODProjector.py:154-174)
    eye = np.eye(r)
    assert np.linalg.norm(eye - phi.T @ Mphi) / np.linalg.norm(eye) < 1e-8
    assert np.linalg.norm(M @ phi - Mphi) / np.linalg.norm(Mphi) < 1e-8


def test_pod_methods_agree_with_each_other(golden_dir):
    """SURVEY section 8c cross-check (3): hep / ghep / inverse_ghep give the same spectrum."""
    g = _load(golden_dir, "pod_from_data.npz")
    for s in (0, 1):
        ref = g["d_hep_%d" % s]
        for m in ("ghep", "inverse_ghep"):
            np.testing.assert_allclose(g["d_%s_%d" % (m, s)][:8], ref[:8], rtol=1e-8)


def test_weighted_l2_norm(golden_dir):
    g = _load(golden_dir, "pod_from_data.npz")
    M = _csr(g, int(g["N"]))
    np.testing.assert_allclose(hf_o.weighted_l2_norm_vector(g["wl2_in"], M), g["wl2_out"], rtol=1e-14)


def test_mean_jtj_matches_reference(golden_dir):
    g = _load(golden_dir, "mean_jtj.npz")
    J, x = g["J"], g["x"]
    for j in range(x.shape[1]):
        np.testing.assert_allclose(hf_o.mean_jtj_mult(J, x[:, j]), g["y"][:, j], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(hf_o.mean_jtj_mult(J, x[:, j], g["Gamma_inv"]), g["y_gamma"][:, j],
                                   rtol=1e-13, atol=1e-12)
    # block form == column-by-column reference (SURVEY section 8c cross-check (2))
    np.testing.assert_allclose(hf_o.mean_jtj_block(J, x), g["y"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(hf_o.mean_jtj_block(J, x, g["Gamma_inv"]), g["y_gamma"], rtol=1e-13, atol=1e-12)


def test_dense_jtj_jjt_summed(golden_dir):
    g = _load(golden_dir, "operators.npz")
    A, Js, x13, x9 = g["A"], g["Js"], g["x13"], g["x9"]
    y = np.zeros(9)
    hp_o.DenseOperator(A).mult(x13, y)
    np.testing.assert_allclose(y, g["np_mult"], rtol=1e-14)
    np.testing.assert_allclose(A.T @ x9, g["np_transpmult"], rtol=1e-14)
    np.testing.assert_allclose(Js[0].T @ (Js[0] @ x13), g["jtj"], rtol=1e-13)
    np.testing.assert_allclose(Js[0] @ (Js[0].T @ x9), g["jjt"], rtol=1e-13)
    ops = [hp_o.DenseOperator(Ji.T @ Ji) for Ji in Js]
    ys = np.zeros(13)
    hf_o.SummedListOperator(ops, average=True).mult(x13, ys)
    np.testing.assert_allclose(ys, g["summed_avg"], rtol=1e-12)
    # the same mean through the stacked (ndata, q, N) form the device path uses
    np.testing.assert_allclose(hf_o.mean_jtj_mult(Js, x13), g["summed_avg"], rtol=1e-12)
    np.testing.assert_allclose(hf_o.mean_jjt_block(Js, x9[:, None])[:, 0],
                               np.mean([Ji @ (Ji.T @ x9) for Ji in Js], axis=0), rtol=1e-12)


def test_collective_semantics(golden_dir):
    g = _load(golden_dir, "collectives.npz")
    parts = g["parts"]
    P = parts.shape[0]
    for op in ("sum", "avg", "Avg"):
        np.testing.assert_allclose(hf_o.all_reduce(list(parts[:, :, 0]), op), g["array_" + op], rtol=1e-14)
        scal = hf_o.all_reduce([np.array([0.5])] + [np.array([float(p)]) for p in range(1, P)], op)[0]
        np.testing.assert_allclose(scal, g["scalar_" + op], rtol=1e-14)
        np.testing.assert_allclose(hf_o.all_reduce(list(parts[:, :, 0] * 2.0), op), g["collop_" + op], rtol=1e-14)
        np.testing.assert_allclose(hf_o.all_reduce(list(parts), op), g["mmcollop_" + op], rtol=1e-14)
    assert int(g["null_size"]) == 1 and int(g["null_rank"]) == 0
    np.testing.assert_array_equal(g["null_allreduce"], parts[0, :, 0])
    assert int(g["null_bad_op_raises"]) == 1 and int(g["mpi_bad_type_raises"]) == 1
    with pytest.raises(NotImplementedError):
        hf_o.all_reduce([np.zeros(2)], "max")


def test_mass_preconditioned_covariance_and_consumers(golden_dir):
    g = _load(golden_dir, "kle_and_consumers.npz")
    N = g["x"].shape[0]
    M = _csr(g, N)
    y = np.zeros(N)
    hf_o.MassPreconditionedCovarianceOperator(hp_o.DenseOperator(g["Cov"]), hp_o.SparseOperator(M)).mult(g["x"], y)
    np.testing.assert_allclose(y, g["mcm"], rtol=1e-13)
    U, V, s = g["U"], g["V"], g["s"]
    np.testing.assert_allclose(U @ (U.T @ (M @ g["x"])), g["prior_precond_proj"], rtol=1e-12)
    np.testing.assert_allclose(U @ (s * (V.T @ g["x13"])), g["lowrank_mult"], rtol=1e-12)
    np.testing.assert_allclose(V @ (s * (U.T @ g["x"])), g["lowrank_transpmult"], rtol=1e-12)


@pytest.mark.parametrize("shifted", [True, False])
@pytest.mark.parametrize("method", ["hep", "ghep", "inverse_ghep"])
def test_pod_from_data_with_320_snapshots_matches_reference(golden_dir, method, shifted):
    """The same pin on a snapshot set beyond 256 (tests/golden/make_pod_large_golden.py: the reference's construct_subspace on 320
    integer-valued snapshots): the oracle here, the whole-GPU eigensolver path in tests/test_gpu_eig_blocked.py."""
    g = _load(golden_dir, "pod_from_data_320.npz")
    N, r = int(g["N"]), int(g["r"])
    M = _csr(g, N)
    u_data = g["u_int16"].astype(np.float64) * float(g["scale"])
    tag = "%s_%d" % (method, int(shifted))
    d, phi, Mphi, shift = hf_o.pod_from_data(u_data.copy(), M, r, shifted=shifted, method=method)
    np.testing.assert_allclose(shift, g["shift_" + tag], rtol=0, atol=1e-14)
    np.testing.assert_allclose(d, g["d_" + tag], rtol=1e-8)
    cosines = np.abs(np.einsum("ij,ij->j", phi[:, :6], M @ g["phi_" + tag][:, :6]))
    np.testing.assert_allclose(cosines, 1.0, atol=1e-8)


def test_pod_from_data_with_8300_snapshots_matches_reference(golden_dir):
    """More than 8192 snapshots, slowly decaying spectrum (tests/golden/make_pod_huge_golden.py: the reference's construct_subspace,
    'hep', shifted, on 8300 x 600 integer-valued snapshots rebuilt from the stored seed): the oracle here (one la.eigh of 8300 x 8300,
    about a minute), the device's exact n x n route in tests/test_gpu_eig_blocked.py."""
    import sys
    sys.path.insert(0, golden_dir)
    from make_pod_huge_golden import snapshots
    g = _load(golden_dir, "pod_from_data_8300.npz")
    N, r = int(g["N"]), int(g["r"])
    M = _csr(g, N)
    u_data = snapshots(int(g["seed"]), int(g["n"]), N, int(g["K"]))
    d, phi, Mphi, shift = hf_o.pod_from_data(u_data.copy(), M, r, shifted=True, method="hep")
    np.testing.assert_allclose(shift, g["shift_hep_1"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(d, g["d_hep_1"], rtol=1e-8)
    cosines = np.abs(np.einsum("ij,ij->j", phi[:, :r - 2], M @ g["phi_hep_1"][:, :r - 2]))
    np.testing.assert_allclose(cosines, 1.0, atol=1e-8)
