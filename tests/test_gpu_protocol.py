"""Drop-in boundary: the projectors driven through the REFERENCE'S OWN observable / prior / vector protocol.

``tests/golden/protocol.npz`` was made by running the reference's ``PODProjector``, ``ActiveSubspaceProjector`` (batched and
serialized, prior-preconditioned or not, input and output) and ``KLEProjector`` over a numpy PDE wrapped in the reference's
``LinearStateObservable`` (tests/golden/make_goldens.py section 7; hp.doublePass[G] = oracle/hippylib_restated.py run over the
reference's operator objects, recording every block).  Here the same PDE is wrapped in ``fake_pde.ProtocolObservable`` --
a class with the reference observable's methods and NOTHING else (no ``sample_observables``, ``jacobian_data`` ... hooks) -- and
handed to ``hippyflow_amd``'s projectors, exactly as an existing driver would.  Checked: the snapshots / operator actions the
device path computes from it (1e-12 of the reference's), the probe block (1e-13), the eigenvalues (1e-9) and the
encoder / decoder subspaces."""
import os
import sys

import numpy as np
import pytest

import hippyflow_amd as hf

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
import fake_pde as fp  # noqa: E402

pytestmark = pytest.mark.gpu

HOOKS = ("sample_observables", "observable_stream", "jacobian_data", "jacobian_stream", "jtj_host_operator", "jacobian")


@pytest.fixture(scope="module")
def ctx():
    if hf.device_count() < 1:
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    return hf.Context.default()


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "protocol.npz"))


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def same_columns_up_to_sign(A, B, tol):
    for j in range(B.shape[1]):
        assert min(rel(A[:, j], B[:, j]), rel(-A[:, j], B[:, j])) < tol, j


def setup(g, **prior_kw):
    n, q = int(g["n"]), int(g["q"])
    obs = fp.ProtocolObservable(fp.NumpyProblem(n, hf.HostVector), fp.MatrixOperator(fp.observation_matrix(q, n)))
    assert not any(hasattr(obs, h) for h in HOOKS)
    prior = fp.NumpyPrior(n, hf.HostVector, **prior_kw)
    assert not hasattr(prior, "sample_block") and not hasattr(prior, "C")
    hf.parRandom.reseed(int(g["seed"]))
    hf.parRandom.split(0)
    return obs, prior


def test_pod_projector_runs_the_reference_sampling_loop(ctx, g, tmp_path):
    obs, prior = setup(g)
    params = hf.PODParameterList()
    params['sample_per_process'], params['rank'], params['oversampling'] = 4 * int(g["n_samples"]), int(g["rank"]), int(g["oversampling"])
    params['output_directory'], params['verbose'] = str(tmp_path) + "/", False
    pod = hf.PODProjector(obs, prior, parameters=params)
    pod.construct_subspace()
    assert obs.n_fwd_solve == params['sample_per_process'] and prior.n_samples == params['sample_per_process']
    np.testing.assert_allclose(pod.LocalObservables.to_vectors(), g["pod_snapshots"], rtol=1e-12, atol=1e-14)
    # the operator the reference built from ITS snapshots, applied to the reference's own probe, against ours
    Y = hf.MultiVector(int(g["q"]), pod.parameters['rank'] + pod.parameters['oversampling'])
    hf.SnapshotGramOperator(pod.LocalObservables).matMvMult(hf.MultiVector.from_dense(g["pod_call0_app0_in"]), Y)
    assert rel(Y.to_dense(), g["pod_call0_app0_out"]) < 1e-12
    np.testing.assert_allclose(pod.d, g["pod_call0_d"], rtol=1e-9)
    same_columns_up_to_sign(pod.U_MV.to_dense()[:, :3], g["pod_call0_U"][:, :3], 1e-7)
    np.testing.assert_allclose(np.load(str(tmp_path) + "/POD_d.npy"), g["pod_saved_d"], rtol=1e-9)
    # the projection-error test draws its samples through the same loop
    avg, std = pod.test_output_errors(ranks=[2, 5])
    assert avg[1] < avg[0] < 1.0
    # input-output error test (PODProjector.py:541-655) with a prior-preconditioned input basis: against a plain numpy
    # evaluation of the same definition on the same draws
    import scipy.sparse.linalg as spla
    n, ns = int(g["n"]), params['sample_per_process']
    rng = np.random.default_rng(4)
    Vd = rng.standard_normal((n, 8))
    Rm = prior.Rmat.toarray()
    Lc = np.linalg.cholesky(Vd.T @ Rm @ Vd)
    Vd = Vd @ np.linalg.inv(Lc).T                                    # V^T R V = I
    V = hf.MultiVector.from_dense(Vd)
    hf.parRandom.reseed(77)
    pairs = [(2, 2), (8, 6)]
    n0 = obs.n_fwd_solve
    avg, std = pod.input_output_error_test(V, Cinv=prior.R, rank_pairs=pairs)
    assert obs.n_fwd_solve - n0 == ns * (1 + len(pairs))              # the reference's solve count
    from oracle import philox
    U = pod.U_MV.to_dense()
    Bobs, prob = obs.B.A, obs.problem

    def q_of(mvec):
        return Bobs @ spla.spsolve(prob._operator(mvec), prob.f)

    expect = []
    ms = [prior.mean.get_local() + prior._Alu.solve(np.sqrt(prior.Ml) * philox.randn_block(n, 1, 77 | (1 << 32), i)[:, 0]) for i in range(ns)]
    for s_in, r_out in pairs:
        rel_errs = []
        for mi in ms:
            q = q_of(mi)
            qr = q_of(Vd[:, :s_in] @ (Vd[:, :s_in].T @ (Rm @ mi)))
            rel_errs.append(np.linalg.norm(q - U[:, :r_out] @ (U[:, :r_out].T @ qr)) / np.linalg.norm(q))
        expect.append((np.mean(rel_errs), np.std(rel_errs)))
    np.testing.assert_allclose(avg, [e[0] for e in expect], rtol=1e-9)
    np.testing.assert_allclose(std, [e[1] for e in expect], rtol=1e-7)


def test_observable_jacobian_is_the_reference_chain(ctx, g):
    obs, prior = setup(g)
    u, m = obs.generate_vector(hf.STATE), obs.generate_vector(hf.PARAMETER)
    m.set_local(g["prior_draws"][0])
    obs.solveFwd(u, [u, m, None])
    obs.setLinearizationPoint([u, m, None])
    J = hf.ObservableJacobian(obs)
    assert tuple(J.shape) == tuple(g["jac_shape"])

    def hv(a):
        v = hf.HostVector()
        v.init(len(a))
        v.set_local(a)
        return v

    yq, yn, yj = hv(np.zeros(J.shape[0])), hv(np.zeros(J.shape[1])), hv(np.zeros(J.shape[1]))
    J.mult(hv(g["jac_x"]), yq)
    J.transpmult(hv(g["jac_xq"]), yn)
    hf.JTJ(J).mult(hv(g["jac_x"]), yj)                    # host vectors in, host vectors out: nothing touches the device
    np.testing.assert_allclose(yq.get_local(), g["jac_mult"], rtol=1e-12)
    np.testing.assert_allclose(yn.get_local(), g["jac_transpmult"], rtol=1e-12)
    np.testing.assert_allclose(yj.get_local(), g["jac_jtj"], rtol=1e-12)
    assert rel(J.rows(), g["jac_dense"]) < 1e-12
    assert J.ncalls == 4 + J.shape[0]
    # the same object behind the device solve: vectors go through set_local / get_local
    W = np.random.default_rng(0).standard_normal((J.shape[1], 3))
    Y = hf.MultiVector(J.shape[1], 3)
    hf.as_device_operator(hf.JTJ(J), ctx=ctx).matMvMult(hf.MultiVector.from_dense(W), Y)
    assert rel(Y.to_dense(), g["jac_dense"].T @ (g["jac_dense"] @ W)) < 1e-12


def as_parameters(g, serialized, ms_given=False):
    params = hf.ActiveSubspaceParameterList()
    params['samples_per_process'], params['rank'], params['oversampling'] = int(g["n_samples"]), int(g["rank"]), int(g["oversampling"])
    params['serialized_sampling'], params['ms_given'] = serialized, ms_given
    params['verbose'], params['save_and_plot'], params['store_Omega'] = False, False, True
    return params


CASES = [("as_batched_prior", False, True, "input", False), ("as_batched_plain", False, False, "input", False),
         ("as_serial_prior", True, True, "input", False), ("as_serial_plain", True, False, "input", False),
         ("as_batched_output", False, False, "output", False), ("as_serial_output", True, False, "output", False),
         ("as_serial_given", True, True, "input", True)]


@pytest.mark.parametrize("tag,serialized,prior_preconditioned,which,ms_given", CASES)
def test_active_subspace_projector_over_a_reference_observable(ctx, g, tag, serialized, prior_preconditioned, which, ms_given):
    operation = 'JTJ' if which == "input" else 'JJT'
    ns = int(g["n_samples"])

    def projector():
        obs, prior = setup(g)
        AS = hf.ActiveSubspaceProjector(obs, prior, parameters=as_parameters(g, serialized, ms_given))
        if ms_given:
            AS.ms = []
            for row in g["prior_draws"][:ns]:
                v = obs.generate_vector(hf.PARAMETER)
                v.set_local(row)
                AS.ms.append(v)
            AS.zs = ns * [None]
        return AS, obs, prior

    # (a) the first application of the sample-averaged operator, on the reference's own probe block
    AS, obs, prior = projector()
    op = AS._local_operator(operation)
    X = hf.MultiVector.from_dense(g[tag + "_call0_app0_in"])
    Y = hf.MultiVector(X.size(), X.nvec())
    op.matMvMult(X, Y)
    assert rel(Y.to_dense(), g[tag + "_call0_app0_out"]) < 1e-12
    assert obs.n_fwd_solve == ns and (ms_given or prior.n_samples == ns)

    # (b) the whole construction from a fresh generator state
    AS, obs, prior = projector()
    if which == "input":
        d, decoder, encoder = AS.construct_input_subspace(prior_preconditioned=prior_preconditioned)
        Omega = AS.Omega_GN
    else:
        d, decoder, encoder = AS.construct_output_subspace()
        Omega = AS.Omega_NG
    np.testing.assert_allclose(Omega.to_dense(), g[tag + "_call0_Omega"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(d, g[tag + "_call0_d"], rtol=1e-9)
    same_columns_up_to_sign(decoder.to_dense()[:, :3], g[tag + "_call0_U"][:, :3], 1e-7)
    if which == "input":
        same_columns_up_to_sign(encoder.to_dense()[:, :3], g[tag + "_encoder"][:, :3], 1e-7)
        assert AS.prior_preconditioned == prior_preconditioned
    # serialized sampling linearises afresh in both passes, batched once (activeSubspaceProjector.py:163-248 vs :347-397)
    assert obs.n_fwd_solve == (2 * ns if serialized else ns)


def test_serialized_operator_column_route_and_retries(ctx, g):
    """q > 2k sends the probe block through the reference's column loop on host vectors; a failing forward solve is
    answered with a fresh draw."""
    tag, ns = "as_serial_plain", int(g["n_samples"])
    obs, prior = setup(g)
    noise = hf.new_host_vector()
    prior.init_vector(noise, "noise")
    X = hf.MultiVector.from_dense(g[tag + "_call0_app0_in"])
    Y = hf.MultiVector(X.size(), X.nvec())
    op = hf.SeriallySampledJacobianOperator(obs, noise, prior, operation='JTJ', nsamples=ns, materialize=False)
    op.matMvMult(X, Y)
    assert rel(Y.to_dense(), g[tag + "_call0_app0_out"]) < 1e-12
    assert obs.n_inc_solve == 2 * ns * X.nvec()                    # two incremental solves per (sample, column)
    obs, prior = setup(g)
    obs.problem.fail_every = 3
    op = hf.SeriallySampledJacobianOperator(obs, noise, prior, operation='JTJ', nsamples=ns)
    Y.zero()
    op.matMvMult(X, Y)
    assert op.solver_failures == obs.problem.n_fwd // 3 and op.samples_linearized == ns
    assert np.isfinite(Y.to_dense()).all() and rel(Y.to_dense(), g[tag + "_call0_app0_out"]) > 1e-3   # other draws, same law


@pytest.mark.parametrize("export_csr,solver_shapes", [(False, True), (False, False), (True, True)])
@pytest.mark.parametrize("orthogonality", ["mass", "identity"])
def test_kle_projector_over_a_reference_prior(ctx, g, export_csr, solver_shapes, orthogonality):
    """prior.M as a host operator (size from its init_vector, Msolver shaped by M when it cannot shape vectors itself)
    or as a PETSc-like matrix exporting CSR (then M and M^-1 run on the device)."""
    obs, prior = setup(g, export_csr=export_csr, mass_solver_shapes_vectors=solver_shapes)
    params = hf.KLEParameterList()
    params['rank'], params['oversampling'], params['verbose'], params['save_and_plot'] = int(g["rank"]), int(g["oversampling"]), False, False
    kle = hf.KLEProjector(prior, parameters=params)
    assert kle.N == int(g["n"]) and isinstance(kle.M, hf.CsrOperator) == export_csr
    d, decoder, encoder = kle.construct_input_subspace(orthogonality)
    tag = "kle_%s" % orthogonality
    np.testing.assert_allclose(d, g[tag + "_call0_d"], rtol=1e-9)
    same_columns_up_to_sign(decoder.to_dense()[:, :3], g[tag + "_call0_U"][:, :3], 1e-7)
    same_columns_up_to_sign(encoder.to_dense()[:, :3], g[tag + "_encoder"][:, :3], 1e-7)
    assert prior.Rsolver.calls > 0
    if orthogonality == "mass":
        assert (prior.Msolver.calls == 0) == export_csr           # device PCG replaces the host mass solver only for a CSR M
        avg, std = kle.test_errors(ranks=[2, 6])
        assert avg[1] < avg[0] < 1.0


def test_host_operator_through_the_vector_protocol(ctx, g):
    obs, prior = setup(g)
    n = int(g["n"])
    R = hf.as_device_operator(prior.R)                              # no N: read off prior.R.init_vector
    assert isinstance(R, hf.HostCallbackOperator) and R.mode == "mult" and R.shape == (n, n)
    W = np.random.default_rng(3).standard_normal((n, 4))
    Y = hf.MultiVector(n, 4)
    R.matMvMult(hf.MultiVector.from_dense(W), Y)
    assert rel(Y.to_dense(), prior.Rmat @ W) < 1e-13 and prior.R.calls == 4
    Rinv = hf.as_device_operator(prior.Rsolver)
    assert Rinv.mode == "solve"
    Rinv.matMvMult(Y, Y2 := hf.MultiVector(n, 4))
    assert rel(Y2.to_dense(), W) < 1e-10
    # a solver that cannot shape vectors borrows B's init_vector inside doublePassG
    bare = fp.FactorizedSolver(prior.Rmat, with_init_vector=False)
    Omega = hf.MultiVector(n, 8)
    hf.parRandom.normal(1.0, Omega)
    A = hf.npToDeviceOperator(np.diag(np.linspace(1, 2, n)))
    d1, _ = hf.doublePassG(A, prior.R, bare, Omega, 4)
    d2, _ = hf.doublePassG(A, hf.CsrOperator(prior.Rmat), hf.as_device_operator(prior.Rsolver), Omega, 4)
    np.testing.assert_allclose(d1, d2, rtol=1e-9)
    # errors raised inside the host object surface as themselves
    class Boom(fp.MatrixOperator):
        def mult(self, x, y):
            raise FloatingPointError("assembly failed")
    with pytest.raises(FloatingPointError):
        hf.as_device_operator(Boom(prior.Rmat)).matMvMult(Omega, hf.MultiVector(n, 8))


def test_private_and_shared_random_streams(ctx):
    """Probe blocks come from the shared namespace (identical on every rank), noise vectors from the rank's private one
    (distinct across ranks and from every probe column)."""
    n = 4001
    draws = {}
    for rank in (0, 1):
        hf.parRandom.reseed(5)
        hf.parRandom.split(rank)
        Om = hf.MultiVector(n, 3)
        hf.parRandom.normal(1.0, Om)
        noise = hf.new_host_vector()
        noise.init(n)
        hf.parRandom.normal(1.0, noise)
        dv = hf.Vector()
        dv.init(n)
        hf.parRandom.normal(1.0, dv)
        draws[rank] = (Om.to_dense(), noise.get_local(), dv.get_local())
    hf.parRandom.split(0)
    np.testing.assert_array_equal(draws[0][0], draws[1][0])
    for a, b in ((draws[0][1], draws[1][1]), (draws[0][1], draws[0][0][:, 0]), (draws[1][1], draws[0][0][:, 0]),
                 (draws[0][1], draws[0][2]), (draws[0][2], draws[1][2])):
        assert abs(np.corrcoef(a, b)[0, 1]) < 0.08


def test_training_data_and_low_rank_jacobians_from_a_reference_observable(ctx, g, tmp_path):
    """The producers of the training files, driven by the reference's protocol only: (m, q) pairs in both on-disk forms with
    resume (PODProjector.py:118-297) and the per-sample Jacobian SVDs + their (m, q) pairs (activeSubspaceProjector.py:690-900)."""
    import scipy.sparse.linalg as spla
    n, q = int(g["n"]), int(g["q"])
    out = str(tmp_path) + "/"
    obs, prior = setup(g)
    params = hf.PODParameterList()
    params['data_per_process'], params['output_directory'], params['verbose'] = 3, out, False
    pod = hf.PODProjector(obs, prior, parameters=params)
    assert pod.generate_training_data() == 3
    first = [np.load(out + "data_on_rank_0/m_sample_%d.npy" % i) for i in range(3)]
    params['data_per_process'] = 5
    assert pod.generate_training_data() == 3          # resumes AT the last index found (it may have been half written), as upstream
    f = np.load(out + "mq_on_rank0.npz")
    assert sorted(f.files) == ["m_data", "q_data"] and f["m_data"].shape == (5, n) and f["q_data"].shape == (5, q)
    np.testing.assert_array_equal(f["m_data"][:2], np.stack(first[:2]))
    for mi, qi in zip(f["m_data"], f["q_data"]):
        np.testing.assert_allclose(qi, obs.B.A @ spla.spsolve(obs.problem._operator(mi), obs.problem.f), rtol=1e-12)
    assert pod.generate_training_data(sequential=False) == 5
    ms = np.load(out + "ms_on_rank_0.npy")
    assert ms.shape == (5, n) and np.load(out + "qs_on_rank_0.npy").shape == (5, q)
    assert pod.generate_training_data(sequential=False) == 0                    # nothing left to do

    obs, prior = setup(g)
    asp = hf.ActiveSubspaceParameterList()
    asp['jacobian_data_per_process'], asp['output_directory'], asp['verbose'] = 4, out, False
    AS = hf.ActiveSubspaceProjector(obs, prior, parameters=asp)
    U, sig, V = AS.construct_low_rank_Jacobians()
    assert U.shape == (4, q, q) and sig.shape == (4, q) and V.shape == (4, n, q) and obs.n_fwd_solve == 4
    mq = np.load(out + "mq_on_proc0.npz")
    jf = np.load(out + "J_on_proc0.npz")
    assert sorted(jf.files) == ["U_data", "V_data", "sigma_data"]
    try:                                       # the plot the reference leaves beside it (activeSubspaceProjector.py:880-883)
        import matplotlib  # noqa: F401
        assert os.path.getsize(out + "jacobian_singular_values_%d.pdf" % q) > 1000
    except ImportError:
        pass
    u = obs.generate_vector(hf.STATE)
    m = obs.generate_vector(hf.PARAMETER)
    for i in range(4):
        m.set_local(mq["m_data"][i])
        obs.solveFwd(u, [u, m, None])
        obs.setLinearizationPoint([u, m, None])
        Jd = obs.problem.jacobian_dense(obs.B.A)
        np.testing.assert_allclose(np.sort(sig[i])[::-1], np.linalg.svd(Jd, compute_uv=False), rtol=1e-9)
        assert rel((U[i] * sig[i]) @ V[i].T, Jd) < 1e-9
        np.testing.assert_allclose(mq["q_data"][i], obs.B.A @ u.get_local(), rtol=1e-12)

    # the batched form (activeSubspaceProjector.py:906-1045): the samples the subspace was built from, whole arrays under jacobian_data/
    obs, prior = setup(g)
    asp = hf.ActiveSubspaceParameterList()
    asp['serialized_sampling'], asp['samples_per_process'], asp['rank'], asp['oversampling'] = False, 3, 4, 2
    asp['output_directory'], asp['verbose'], asp['save_and_plot'] = out, False, False
    AS = hf.ActiveSubspaceProjector(obs, prior, parameters=asp)
    AS.construct_input_subspace(prior_preconditioned=False)
    Ub, sb, Vb = AS.construct_low_rank_Jacobians()
    r = min(4, q, n)
    assert Ub.shape == (3, q, r) and sb.shape == (3, r) and Vb.shape == (3, n, r)
    ms_b, qs_b = np.load(out + "jacobian_data/ms_on_proc_0.npy"), np.load(out + "jacobian_data/qs_on_proc_0.npy")
    assert ms_b.shape == (3, n) and qs_b.shape == (3, q)
    np.testing.assert_array_equal(np.load(out + "jacobian_data/sigmas_on_proc_0.npy"), sb)
    for i in range(3):                                  # (m_i, q_i, J_i) belong together: the stored linearisation points
        m.set_local(ms_b[i])
        obs.solveFwd(u, [u, m, None])
        obs.setLinearizationPoint([u, m, None])
        np.testing.assert_allclose(qs_b[i], obs.B.A @ u.get_local(), rtol=1e-12)
        Jd = obs.problem.jacobian_dense(obs.B.A)
        assert rel(Ub[i].T @ Jd @ Vb[i], np.diag(sb[i])) < 1e-9 and np.all(sb[i] <= np.linalg.svd(Jd, compute_uv=False)[:r] * (1 + 1e-12))


def test_control_problem_low_rank_jacobians(ctx, tmp_path):
    """A control problem (state equation with a control z on its right-hand side) driven through the reference's protocol:
    the control is sampled next to the parameter, enters the forward solve, and construct_low_rank_control_Jacobians
    (activeSubspaceProjector.py:682-688) dumps the SVDs of Jz = B A^-1 G per sample beside (m, z, q)."""
    n, q, dz, nd = 40, 6, 4, 3
    out = str(tmp_path) + "/"
    prob = fp.NumpyControlProblem(n, dz, hf.HostVector)
    obs = fp.ProtocolObservable(prob, fp.MatrixOperator(fp.observation_matrix(q, n)))
    prior = fp.NumpyPrior(n, hf.HostVector)
    hf.parRandom.reseed(4)
    hf.parRandom.split(0)
    asp = hf.ActiveSubspaceParameterList()
    asp['jacobian_data_per_process'], asp['output_directory'], asp['verbose'] = nd, out, False
    asp['control_jacobian_rank'], asp['jacobian_rank'] = 3, 5
    AS = hf.ActiveSubspaceProjector(obs, prior, control_distribution=fp.ControlDistribution(dz), parameters=asp)
    Uz, sz, Vz = AS.construct_low_rank_control_Jacobians()
    assert Uz.shape == (nd, q, 3) and sz.shape == (nd, 3) and Vz.shape == (nd, dz, 3) and obs.n_fwd_solve == nd
    jf = np.load(out + "Jz_on_proc0.npz")
    assert sorted(jf.files) == ["Uz_data", "Vz_data", "sigmaz_data"]
    mzq = np.load(out + "mzq_on_proc0.npz")
    assert sorted(mzq.files) == ["m_data", "q_data", "z_data"] and mzq["z_data"].shape == (nd, dz)
    assert not os.path.exists(out + "mq_on_proc0.npz") and not os.path.exists(out + "J_on_proc0.npz")
    u, m, z = obs.generate_vector(hf.STATE), obs.generate_vector(hf.PARAMETER), obs.generate_vector(hf.CONTROL)
    for i in range(nd):
        m.set_local(mzq["m_data"][i])
        z.set_local(mzq["z_data"][i])
        obs.solveFwd(u, [u, m, None, z])
        obs.setLinearizationPoint([u, m, None, z])
        np.testing.assert_allclose(mzq["q_data"][i], obs.B.A @ u.get_local(), rtol=1e-12)
        Jd = prob.control_jacobian_dense(obs.B.A)
        sv = np.linalg.svd(Jd, compute_uv=False)
        # rank 3 of 4 from a 3-column probe (no oversampling, as upstream): a Rayleigh-Ritz approximation -- orthonormal factors,
        # U^T Jz V = diag(sigma) by construction, every sigma_i below the true one and the leading one close to it
        assert rel(Uz[i].T @ Uz[i], np.eye(3)) < 1e-10 and rel(Vz[i].T @ Vz[i], np.eye(3)) < 1e-10
        assert rel(Uz[i].T @ Jd @ Vz[i], np.diag(sz[i])) < 1e-9
        assert np.all(sz[i] <= sv[:3] * (1 + 1e-12)) and sz[i][0] > 0.98 * sv[0]
    # the parameter Jacobians of the same control problem: (m, z, q) again, J taken at (m, z)
    U, sig, V = AS.construct_low_rank_Jacobians()
    assert U.shape == (nd, q, 5) and V.shape == (nd, n, 5) and os.path.exists(out + "J_on_proc0.npz")
    mzq = np.load(out + "mzq_on_proc0.npz")
    for i in range(nd):
        m.set_local(mzq["m_data"][i])
        z.set_local(mzq["z_data"][i])
        obs.solveFwd(u, [u, m, None, z])
        obs.setLinearizationPoint([u, m, None, z])
        Jd = prob.jacobian_dense(obs.B.A)
        assert rel(U[i].T @ Jd @ V[i], np.diag(sig[i])) < 1e-9
    # full rank when no control rank is named: then the factorisation is exact
    asp['control_jacobian_rank'] = None
    Uz, sz, Vz = AS.construct_low_rank_control_Jacobians(compress_files=False)
    assert sz.shape == (nd, dz)
    np.testing.assert_allclose(sz[-1], np.linalg.svd((Uz[-1] * sz[-1]) @ Vz[-1].T, compute_uv=False), rtol=1e-10)
    zlast = obs.generate_vector(hf.CONTROL)
    assert zlast.size() == dz
    with pytest.raises(AssertionError):
        hf.ActiveSubspaceProjector(obs, prior, parameters=asp).construct_low_rank_control_Jacobians()
