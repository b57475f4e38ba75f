"""Derivative training data from stored Jacobians and the ``.npz`` datasets the training scripts read
(reference: modeling/dataGenerator.py).

The reference's ``DataGenerator`` is bound to FEniCS PDE solves (the Jacobian actions are adjoint/incremental solves,
dataGenerator.py:120-260) and stays a host black box.  What follows the solve is on the device path: with the
per-sample Jacobians ``J_i`` (q x N) stored as blocks (one vector per Jacobian row, the layout of
``MeanJTJfromDataOperator``, operatorWrappers.py:62-64), the three derivative products the generator dumps are
tall-skinny contractions over the same blocks:

* ``JstarPhi_i = J_i^T (M Phi)``  (N x r_out; dataGenerator.py:163-169)  -- ``MvDSmatMult`` per sample,
* ``JPsi_i = J_i Psi``            (q x r_in;  dataGenerator.py:170-176)  -- ONE ``dot_mv`` over all samples,
* ``U_i, sigma_i, V_i = accuracyEnhancedSVD(J_i, Omega, rM, s=1)``       (dataGenerator.py:178-191),

and ``compress_dataset`` (dataGenerator.py:520-660) stacks them into ``.npz`` files whose key names are the contract
with the training side:

    mq_data.npz        m_data (n, dM), q_data (n, dQ)            [mzq_data.npz adds z_data]
    JstarPhi_data.npz  JstarPhi_data (n, dM, r), Phi, MPhi
    JPsi_data.npz      JPsi_data (n, dQ, r), Psi, input_encoder
    Jsvd_data.npz      U_data (n, dQ, rM), sigma_data (n, rM), V_data (n, dM, rM)
"""
import os

import numpy as np

from .multivector import MultiVector, MvDSmatMult
from .operators import DenseJacobianOperator
from .randomized import accuracyEnhancedSVD, parRandom


def _as_jacobian_block(J, ctx=None):
    """(ndata, q, N) array or a block of ndata*q vectors -> (block, ndata, q)."""
    if isinstance(J, tuple):
        blk, ndata, q = J
        assert blk.nvec() == ndata * q
        return blk, int(ndata), int(q)
    J = np.asarray(J, dtype=np.float64)
    assert J.ndim == 3, "stored Jacobians must have shape (ndata, q, N)"
    ndata, q, N = J.shape
    return MultiVector.from_vectors(J.reshape(ndata * q, N), ctx=ctx), ndata, q


def jacobian_times_input_basis(J, Psi, ctx=None):
    """JPsi_data[i] = J_i Psi for every sample in one contraction (dataGenerator.py:170-176).  ``Psi``: (N, r)."""
    blk, ndata, q = _as_jacobian_block(J, ctx)
    P = Psi if isinstance(Psi, MultiVector) else MultiVector.from_dense(np.asarray(Psi, dtype=np.float64), ctx=blk.ctx)
    out = blk.dot_mv(P)                                   # (ndata*q, r): reduction over N on the device
    return out.reshape(ndata, q, P.nvec())


def jacobian_transpose_times_output_basis(J, MPhi, ctx=None):
    """JstarPhi_data[i] = J_i^T (M Phi) for every sample (dataGenerator.py:163-169).  ``MPhi``: (q, r) host array."""
    blk, ndata, q = _as_jacobian_block(J, ctx)
    MPhi = np.ascontiguousarray(np.asarray(MPhi, dtype=np.float64))
    assert MPhi.shape[0] == q
    r = MPhi.shape[1]
    N = blk.size()
    out = np.empty((ndata, N, r))
    Y = MultiVector(N, r, ctx=blk.ctx)
    for i in range(ndata):
        Ji = blk.view(i * q, q)                           # the q rows of sample i, no copy
        MvDSmatMult(Ji, MPhi, Y)                          # (N x q)(q x r)
        out[i] = Y.to_dense()
    return out


def jacobian_svds(J, rank, oversampling=0, seed=0, s=1, ctx=None):
    """Randomized SVD of every stored Jacobian with hippylib's accuracyEnhancedSVD (dataGenerator.py:178-191):
    returns U_data (ndata, q, rank), sigma_data (ndata, rank), V_data (ndata, N, rank).  As in the reference a
    fresh Gaussian Omega (N x rank) is drawn for every sample."""
    blk, ndata, q = _as_jacobian_block(J, ctx)
    N = blk.size()
    k = rank + oversampling
    assert k <= q, "rank + oversampling cannot exceed the number of Jacobian rows"
    U_data, s_data, V_data = np.empty((ndata, q, rank)), np.empty((ndata, rank)), np.empty((ndata, N, rank))
    rng = parRandom if seed == 0 else type(parRandom)(seed)   # seed 0: the process-wide stream, as hp.parRandom
    Omega = MultiVector(N, k, ctx=blk.ctx)
    for i in range(ndata):
        op = DenseJacobianOperator(blk.view(i * q, q))
        rng.normal(1.0, Omega)
        U, sig, V = accuracyEnhancedSVD(op, Omega, rank, s=s)
        U_data[i], s_data[i], V_data[i] = U.to_dense(), sig, V.to_dense()
    return U_data, s_data, V_data


# ------------------------------------------------------------------ .npz writers / readers (key names = the contract)
def save_mq_data(file_path, m_data, q_data, z_data=None):
    """mq_data.npz / mzq_data.npz (dataGenerator.py:634-638)."""
    os.makedirs(file_path, exist_ok=True)
    if z_data is not None:
        np.savez_compressed(os.path.join(file_path, "mzq_data.npz"), m_data=m_data, q_data=q_data, z_data=z_data)
    else:
        np.savez_compressed(os.path.join(file_path, "mq_data.npz"), m_data=m_data, q_data=q_data)


def save_JstarPhi_data(file_path, JstarPhi_data, output_decoder, output_encoder):
    """JstarPhi_data.npz (dataGenerator.py:643)."""
    os.makedirs(file_path, exist_ok=True)
    np.savez_compressed(os.path.join(file_path, "JstarPhi_data.npz"), JstarPhi_data=JstarPhi_data, Phi=output_decoder,
                        MPhi=output_encoder)


def save_JPsi_data(file_path, JPsi_data, input_decoder, input_encoder):
    """JPsi_data.npz (dataGenerator.py:645)."""
    os.makedirs(file_path, exist_ok=True)
    np.savez_compressed(os.path.join(file_path, "JPsi_data.npz"), JPsi_data=JPsi_data, Psi=input_decoder,
                        input_encoder=input_encoder)


def save_Jsvd_data(file_path, U_data, sigma_data, V_data):
    """Jsvd_data.npz (dataGenerator.py:647)."""
    os.makedirs(file_path, exist_ok=True)
    np.savez_compressed(os.path.join(file_path, "Jsvd_data.npz"), U_data=U_data, sigma_data=sigma_data, V_data=V_data)


def derivative_dataset(file_path, J, m_data=None, q_data=None, output_decoder=None, output_encoder=None,
                       input_decoder=None, input_encoder=None, svd_rank=None, seed=0, ctx=None):
    """The post-solve half of DataGenerator.generate + compress_dataset for stored Jacobians: writes whichever of
    JstarPhi_data.npz / JPsi_data.npz / Jsvd_data.npz the arguments select (same precedence as
    dataGenerator.py:163-191: output basis, else input basis, else randomized SVD) plus mq_data.npz when the
    samples are given.  Returns the dict of arrays written."""
    blk = _as_jacobian_block(J, ctx)
    out = {}
    if m_data is not None and q_data is not None:
        save_mq_data(file_path, m_data, q_data)
        out.update(m_data=m_data, q_data=q_data)
    if output_decoder is not None:
        assert output_encoder is not None, "the output encoder M Phi is what the Jacobian transpose acts on"
        out["JstarPhi_data"] = jacobian_transpose_times_output_basis(blk, output_encoder)
        save_JstarPhi_data(file_path, out["JstarPhi_data"], output_decoder, output_encoder)
    elif input_decoder is not None:
        out["JPsi_data"] = jacobian_times_input_basis(blk, input_decoder)
        save_JPsi_data(file_path, out["JPsi_data"], input_decoder, input_encoder)
    else:
        assert svd_rank is not None, "pass an output basis, an input basis or svd_rank"
        out["U_data"], out["sigma_data"], out["V_data"] = jacobian_svds(blk, svd_rank, seed=seed)
        save_Jsvd_data(file_path, out["U_data"], out["sigma_data"], out["V_data"])
    return out


# ------------------------------------------------------------------ (m, q) training pairs from the host PDE loop
def generate_training_data(observable, prior, n_data, output_directory, noise=None, control_distribution=None, rank=0,
                           check_for_data=True, sequential=True, compress_files=True, max_solver_retries=100, u_init=None):
    """PODProjector.generate_training_data (modeling/PODProjector.py:118-297): ``n_data`` (parameter, observable) pairs of this
    rank, drawn and solved by the reference's loop (noise -> ``prior.sample`` -> ``observable.solveFwd`` -> ``evalu``; a failing
    solve is answered with a fresh draw), written in the reference's two on-disk forms and RESUMED from what is already there:

    * ``sequential``: one ``m_sample_<i>.npy`` / ``q_sample_<i>.npy`` (+ ``z_sample_<i>.npy``) per pair under
      ``data_on_rank_<rank>/``, then (``compress_files``) ``mq_on_rank<rank>.npz`` / ``mqz_on_rank<rank>.npz`` with keys
      ``m_data``, ``q_data`` (, ``z_data``);
    * otherwise growing arrays ``ms_on_rank_<rank>.npy`` / ``qs_on_rank_<rank>.npy`` (/ ``zs_...``), rewritten after every pair.

    Nothing here touches the GPU: it is the producer side of the file formats the training scripts read.  Returns the
    number of pairs generated in this call."""
    from . import hostvec as H
    os.makedirs(output_directory, exist_ok=True)
    if noise is None:
        noise = H.new_host_vector(observable.mpi_comm() if hasattr(observable, "mpi_comm") else None)
        prior.init_vector(noise, "noise")
    u, m = observable.generate_vector(H.STATE), observable.generate_vector(H.PARAMETER)
    z = None if control_distribution is None else observable.generate_vector(H.CONTROL)
    names = ("m", "q") if z is None else ("m", "q", "z")

    def one_pair():
        last = None
        for _ in range(max_solver_retries + 1):
            m.zero()
            noise.zero()
            parRandom.normal(1, noise)
            prior.sample(noise, m)
            point = [u, m, None]
            if z is not None:
                z.zero()
                control_distribution.sample(z)
                point.append(z)
            if u_init is not None:               # the non-sequential form restarts every solve from the state at the mean (:268-269)
                u.zero()
                u.axpy(1.0, u_init)
            try:
                observable.solveFwd(u, point)
            except Exception as exc:             # noqa: BLE001 -- "Issue with the forward solution, moving on." (:218)
                last = exc
                continue
            pair = {"m": m.get_local(), "q": observable.evalu(u).get_local()}
            if z is not None:
                pair["z"] = z.get_local()
            return pair
        raise RuntimeError("forward solve failed for %d consecutive draws (last error: %r)" % (max_solver_retries + 1, last)) from last

    made = 0
    if sequential:
        folder = os.path.join(output_directory, "data_on_rank_%d" % rank) + "/"
        os.makedirs(folder, exist_ok=True)
        done = 0
        if check_for_data:                       # resume after the last index every kind of file has reached (:143-181)
            have = {n: [int(f[len(n) + 8:-4]) for f in os.listdir(folder) if f.startswith(n + "_sample_") and f.endswith(".npy")] for n in names}
            if all(have[n] for n in names):
                done = min(max(v) for v in have.values())
        for i in range(done, n_data):
            pair = one_pair()
            for n in names:
                np.save(folder + "%s_sample_%d.npy" % (n, i), pair[n])
            made += 1
        if compress_files:
            stacked = {n + "_data": np.stack([np.load(folder + "%s_sample_%d.npy" % (n, i)) for i in range(n_data)]) for n in names}
            np.savez_compressed(os.path.join(output_directory, ("mq_on_rank%d.npz" if z is None else "mqz_on_rank%d.npz") % rank), **stacked)
        return made
    paths = {n: os.path.join(output_directory, "%ss_on_rank_%d.npy" % (n, rank)) for n in names}
    rows = {n: [] for n in names}
    if check_for_data and all(os.path.isfile(p) for p in paths.values()):
        loaded = {n: np.load(paths[n]) for n in names}
        keep = min(a.shape[0] for a in loaded.values())
        rows = {n: list(loaded[n][:keep]) for n in names}
    for _ in range(len(rows["m"]), n_data):
        pair = one_pair()
        for n in names:
            rows[n].append(pair[n])
            np.save(paths[n], np.array(rows[n]))
        made += 1
    return made
