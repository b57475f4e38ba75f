"""CPU suite: the host side of the reference's protocols (no GPU, no device call): dolfin-like host vectors, the
``ObservableJacobian`` / ``JTJ`` / ``JJT`` chain on them against the reference-generated fixture, and the communicator
helpers of collectives/comm_utils.py:19-75."""
import os
import sys

import numpy as np
import pytest

import hippyflow_amd as hf
from hippyflow_amd import hostvec as H

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
import fake_pde as fp  # noqa: E402


def hv(a):
    v = hf.HostVector()
    v.init(len(a))
    v.set_local(a)
    return v


def test_host_vector_surface():
    x, y = hv([1.0, 2.0, 3.0]), hv([0.5, 0.5, 0.5])
    assert x.size() == x.local_size() == len(x) == 3 and x.mpi_comm().Get_size() == 1
    x.axpy(2.0, y)
    np.testing.assert_array_equal(x.get_local(), [2.0, 3.0, 4.0])
    assert x.inner(y) == 4.5 and abs(x.norm("l2") - np.sqrt(29.0)) < 1e-15 and x.norm("linf") == 4.0
    x *= 0.5
    c = hf.HostVector(x)                       # dl.Vector(other): a copy
    x.zero()
    np.testing.assert_array_equal(c.get_local(), [1.0, 1.5, 2.0])
    np.testing.assert_array_equal((c * y).get_local(), [0.5, 0.75, 1.0])       # u * indicator (observable.py:59)
    with pytest.raises(ValueError):
        x.set_local([1.0])
    got = c.get_local()
    got[0] = 99.0                              # get_local hands out a copy, as dolfin does
    assert c.get_local()[0] == 1.0
    mv = hf.HostMultiVector(c, 4)
    assert mv.nvec() == 4 and mv[2].size() == 3 and not mv[2].get_local().any()
    mv[1].set_local([1, 2, 3])
    cp = hf.HostMultiVector(mv)
    mv.zero()
    assert cp[1].get_local()[2] == 3.0
    assert H.is_host_vector(c) and not H.is_host_vector(np.zeros(3))
    hf.set_host_vector_factory(lambda comm: "made")
    try:
        assert hf.new_host_vector() == "made"
    finally:
        hf.set_host_vector_factory(None)
    assert isinstance(hf.new_host_vector(), hf.HostVector)        # no FEniCS in this image


def test_shape_lookup_follows_solver2operator():
    class WithOperator:
        def operator(self):
            return fp.MatrixOperator(np.eye(5))

    class WithGetOperator:
        def get_operator(self):
            return fp.MatrixOperator(np.eye(7))

    assert H.shape_with(H.find_init_vector(WithOperator()), 0).size() == 5
    assert H.shape_with(H.find_init_vector(WithGetOperator()), 1).size() == 7
    assert H.find_init_vector(object()) is None
    assert H.shape_with(lambda x: x.init(3), 0).size() == 3          # init_vector(x) without a dim (activeSubspaceProjector.py:144)


def test_jacobian_chain_on_host_vectors_matches_the_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "protocol.npz"))
    n, q = int(g["n"]), int(g["q"])
    obs = fp.ProtocolObservable(fp.NumpyProblem(n, hf.HostVector), fp.MatrixOperator(fp.observation_matrix(q, n)))
    u, m = obs.generate_vector(hf.STATE), obs.generate_vector(hf.PARAMETER)
    m.set_local(g["prior_draws"][0])
    obs.solveFwd(u, [u, m, None])
    obs.setLinearizationPoint([u, m, None])
    J = hf.ObservableJacobian(obs)
    assert tuple(J.shape) == (q, n)
    yq, yn, yj, yjj = hv(np.zeros(q)), hv(np.zeros(n)), hv(np.zeros(n)), hv(np.zeros(q))
    J.mult(hv(g["jac_x"]), yq)
    J.transpmult(hv(g["jac_xq"]), yn)
    jtj = hf.JTJ(J)
    jtj.mult(hv(g["jac_x"]), yj)
    hf.JJT(J).mult(hv(g["jac_xq"]), yjj)
    np.testing.assert_allclose(yq.get_local(), g["jac_mult"], rtol=1e-12)
    np.testing.assert_allclose(yn.get_local(), g["jac_transpmult"], rtol=1e-12)
    np.testing.assert_allclose(yj.get_local(), g["jac_jtj"], rtol=1e-12)
    Jd = g["jac_dense"]
    np.testing.assert_allclose(yjj.get_local(), Jd @ (Jd.T @ g["jac_xq"]), rtol=1e-11)
    np.testing.assert_allclose(J.rows(), Jd, rtol=1e-10, atol=1e-14)
    v = hf.HostVector()
    jtj.init_vector(v, 0)
    assert v.size() == n
    with pytest.raises(ValueError):
        J.init_vector(v, 2)
    # the mean of operators on host vectors (SummedListOperator), and the serialized accumulation on host column lists
    ysum = hv(np.zeros(n))
    hf.SummedListOperator([jtj, jtj], average=True).mult(hv(g["jac_x"]), ysum)
    np.testing.assert_allclose(ysum.get_local(), g["jac_jtj"], rtol=1e-12)
    prior = fp.NumpyPrior(n, hf.HostVector)
    X, Y = hf.HostMultiVector(v, 2), hf.HostMultiVector(v, 2)
    X[0].set_local(g["jac_x"])
    X[1].set_local(np.ones(n))
    given = hv(g["prior_draws"][0])
    op = hf.SeriallySampledJacobianOperator(obs, None, prior, operation='JTJ', ms=[given, given], average=True)
    op.matMvMult(X, Y)
    np.testing.assert_allclose(Y[0].get_local(), g["jac_jtj"], rtol=1e-11)
    np.testing.assert_allclose(Y[1].get_local(), Jd.T @ (Jd @ np.ones(n)), rtol=1e-11)


def test_control_jacobian_chain_on_host_vectors():
    """ObservableControlJacobian (controlJacobian.py:21-95) over a control problem: applyCz -> solveFwdIncremental -> applyB
    and its transpose, against the dense Jz = B A^-1 G of the numpy problem; shapes through init_vector(x, 3)."""
    n, q, dz = 40, 6, 4
    prob = fp.NumpyControlProblem(n, dz, hf.HostVector)
    obs = fp.ProtocolObservable(prob, fp.MatrixOperator(fp.observation_matrix(q, n)))
    assert obs.is_control_problem and len(obs.generate_vector()) == 4
    u, m, z = obs.generate_vector(hf.STATE), obs.generate_vector(hf.PARAMETER), obs.generate_vector(hf.CONTROL)
    rng = np.random.default_rng(0)
    m.set_local(0.3 * rng.standard_normal(n))
    z.set_local(rng.standard_normal(dz))
    x = [u, m, None, z]
    obs.solveFwd(u, x)
    obs.setLinearizationPoint(x)
    Jz = hf.ObservableControlJacobian(obs)
    assert isinstance(Jz, hf.Jacobian) and tuple(Jz.shape) == (q, dz)
    Jd = prob.control_jacobian_dense(obs.B.A)
    xz, xq = rng.standard_normal(dz), rng.standard_normal(q)
    yq, yz = hv(np.zeros(q)), hv(np.zeros(dz))
    Jz.mult(hv(xz), yq)
    Jz.transpmult(hv(xq), yz)
    np.testing.assert_allclose(yq.get_local(), Jd @ xz, rtol=1e-11)
    np.testing.assert_allclose(yz.get_local(), Jd.T @ xq, rtol=1e-11)
    np.testing.assert_allclose(Jz.dense(), Jd, rtol=1e-10, atol=1e-14)         # dz < q: by columns
    np.testing.assert_allclose(Jz.rows(), Jd, rtol=1e-10, atol=1e-14)
    v = hf.HostVector()
    Jz.init_vector(v, 1)
    assert v.size() == dz
    Jz.init_vector(v, 0)
    assert v.size() == q
    # the control enters the forward solve: the parameter Jacobian is taken at (m, z)
    J = hf.ObservableJacobian(obs)
    np.testing.assert_allclose(J.dense(), prob.jacobian_dense(obs.B.A), rtol=1e-10, atol=1e-14)
    with pytest.raises(AssertionError):            # not a control problem: no applyCz
        hf.ObservableControlJacobian(fp.ProtocolObservable(fp.NumpyProblem(n, hf.HostVector), obs.B))
    with pytest.raises(NotImplementedError):
        hf.Jacobian().transpmult(None, None)


class _MpiLikeComm:
    def __init__(self, size=1, rank=0):
        self._size, self._rank, self.splits = size, rank, []

    def Get_size(self):
        return self._size

    def Get_rank(self):
        return self._rank

    def Split(self, color, key):
        self.splits.append((color, key))
        return ("split", color, key)

    def bcast(self, obj, root=0):
        return obj


def test_split_communicators_without_mesh_partitioning(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "1")
    mesh_comm, coll_comm = hf.splitCommunicators(None, 1, 1)
    assert mesh_comm.Get_size() == 1 and mesh_comm.rank == 0 and coll_comm is None
    world = _MpiLikeComm(size=4, rank=3)
    mesh_comm, coll_comm = hf.splitCommunicators(world, 1, 4)
    assert world.splits == [(3, 0), (0, 3)]                       # color / key of comm_utils.py:35-39 for n_subdomain = 1
    with pytest.raises(NotImplementedError):
        hf.splitCommunicators(world, 2, 2)
    with pytest.raises(AssertionError):
        hf.splitCommunicators(world, 1, 3)
    assert hf.checkMeshConsistentPartitioning(object(), hf.NullCollective()) is True
    # a one-rank mpi communicator needs no device communicator at all
    coll = hf.MultipleSerialPDEsCollective(_MpiLikeComm())
    assert isinstance(coll, hf.NullCollective) and coll.size() == 1
    assert hf.MultipleSamePartitioningPDEsCollective(coll) is coll


def test_compress_dataset_archives_and_keys(tmp_path):
    """compress_dataset (dataGenerator.py:495-700) on per-sample files written by hand: which archives appear, their keys, what the
    clean-up removes; no GPU involved."""
    root = str(tmp_path) + "/"
    rng = np.random.default_rng(0)
    nd, dM, dQ, dZ, r = 3, 6, 4, 2, 2
    for folder in ("mzq_data", "J_data", "Jz_data"):
        os.makedirs(root + folder)
    for i in range(nd):
        np.save(root + "mzq_data/m_sample_%d.npy" % i, rng.standard_normal(dM))
        np.save(root + "mzq_data/q_sample_%d.npy" % i, rng.standard_normal(dQ))
        np.save(root + "mzq_data/z_sample_%d.npy" % i, rng.standard_normal(dZ))
        np.save(root + "J_data/JstarPhi%d.npy" % i, rng.standard_normal((dM, r)))
        np.save(root + "J_data/U_sample_%d.npy" % i, rng.standard_normal((dQ, r)))
        np.save(root + "J_data/sigma_sample_%d.npy" % i, rng.standard_normal(r))
        if i < nd - 1:                                   # an incomplete set: not archived
            np.save(root + "J_data/V_sample_%d.npy" % i, rng.standard_normal((dM, r)))
        for stem, shape in (("Uz_sample_%d", (dQ, r)), ("sigmaz_sample_%d", (r,)), ("Vz_sample_%d", (dZ, r))):
            np.save(root + "Jz_data/" + stem % i + ".npy", rng.standard_normal(shape))
    Phi = rng.standard_normal((dQ, r))
    hf.compress_dataset(root, derivatives=(1, 1), clean_up=False, has_z_data=True, output_decoder=Phi, output_encoder=Phi)
    assert sorted(f for f in os.listdir(root) if f.endswith(".npz")) == ["JstarPhi_data.npz", "Jzsvd_data.npz", "mzq_data.npz"]
    f = np.load(root + "JstarPhi_data.npz")
    assert sorted(f.files) == ["JstarPhi_data", "MPhi", "Phi"] and f["JstarPhi_data"].shape == (nd, dM, r)
    np.testing.assert_array_equal(f["JstarPhi_data"][1], np.load(root + "J_data/JstarPhi1.npy"))
    fz = np.load(root + "Jzsvd_data.npz")
    assert sorted(fz.files) == ["Uz_data", "Vz_data", "sigmaz_data"] and fz["Vz_data"].shape == (nd, dZ, r)
    assert np.load(root + "mzq_data.npz")["z_data"].shape == (nd, dZ)
    os.remove(root + "mzq_data.npz")
    hf.compress_dataset(root, derivatives=(1, 1), clean_up=True, has_z_data=True, derivatives_only=True)
    assert not os.path.exists(root + "mzq_data.npz") and os.path.isdir(root + "mzq_data") and not os.path.exists(root + "J_data")
    with pytest.raises(FileNotFoundError):
        hf.compress_dataset(root + "nothing/")
    with pytest.raises(AssertionError):
        hf.compress_dataset(root, derivatives=(0, 1), has_z_data=False)
    assert hf.data_generator_settings()["oversample"] == 10


def test_pde_side_names_are_forwarded_to_an_installed_reference_package(monkeypatch):
    """`from hippyflow import *` -> `from hippyflow_amd import *`: the classes that BUILD the host objects (BiLaplacian2D,
    LinearStateObservable, ...; confusion_problem_setup.py:94, confusion_linear_observable.py:148) are the reference
    package's own, forwarded when it is installed and absent -- with an error that says why -- when it is not."""
    import importlib
    import importlib.machinery
    import types

    assert "BiLaplacian2D" not in hf.__all__ and "PODProjector" in hf.__all__ and "operators" not in hf.__all__
    with pytest.raises(AttributeError, match="forwarded, not re-implemented"):
        hf.BiLaplacian2D
    with pytest.raises(AttributeError):
        hf.no_such_name

    def fake(name, **members):
        mod = types.ModuleType(name)
        mod.__spec__ = importlib.machinery.ModuleSpec(name, loader=None, is_package=True)
        mod.__path__ = []
        mod.__dict__.update(members)
        monkeypatch.setitem(sys.modules, name, mod)
        return mod

    class BiLaplacian2D:            # stands for the reference's FEniCS-side class
        pass

    for name in ("dolfin", "hippylib", "hippyflow", "hippyflow.modeling"):
        fake(name)
    fake("hippyflow.modeling.maternPrior", BiLaplacian2D=BiLaplacian2D)
    try:
        assert hf._reference_package_importable()
        assert hf.BiLaplacian2D is BiLaplacian2D              # first use imports the reference module ...
        assert hf.__dict__["BiLaplacian2D"] is BiLaplacian2D  # ... and the name stays
        with pytest.raises(AttributeError, match="hippyflow.modeling.observable"):
            hf.LinearStateObservable                          # a reference install without that module: says which
        ns = {}
        importlib.reload(hf)
        exec("from hippyflow_amd import *", ns)
        assert ns["BiLaplacian2D"] is BiLaplacian2D and "PODProjector" in ns
    finally:
        hf.__dict__.pop("BiLaplacian2D", None)
        monkeypatch.undo()
        importlib.reload(hf)
    assert "BiLaplacian2D" not in hf.__all__
