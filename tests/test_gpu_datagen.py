"""DataGenerator / compress_dataset (modeling/dataGenerator.py:25-700) over the numpy PDE, through the reference's protocol only:
file names and archive keys are the reference's, the derivative arrays are checked against the dense Jacobians of the numpy
problem re-linearised at the stored samples."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import hippyflow_amd as hf

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
import fake_pde as fp  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    if hf.device_count() < 1:
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    return hf.Context.default()


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def make(n=48, q=7, dz=None):
    prob = fp.NumpyControlProblem(n, dz, hf.HostVector) if dz else fp.NumpyProblem(n, hf.HostVector)
    obs = fp.ProtocolObservable(prob, fp.MatrixOperator(fp.observation_matrix(q, n)))
    prior = fp.NumpyPrior(n, hf.HostVector)
    hf.parRandom.reseed(11)
    hf.parRandom.split(0)
    settings = hf.data_generator_settings()
    settings['verbose'] = False
    return obs, prior, settings


def jacobians_at(obs, m_data, z_data=None):
    u, m = obs.generate_vector(hf.STATE), obs.generate_vector(hf.PARAMETER)
    z = obs.generate_vector(hf.CONTROL) if z_data is not None else None
    for i in range(len(m_data)):
        m.set_local(m_data[i])
        point = [u, m, None]
        if z is not None:
            z.set_local(z_data[i])
            point.append(z)
        obs.solveFwd(u, point)
        obs.setLinearizationPoint(point)
        Jz = obs.problem.control_jacobian_dense(obs.B.A) if z is not None else None
        yield i, obs.B.A @ u.get_local(), obs.problem.jacobian_dense(obs.B.A), Jz


def test_generate_with_randomized_svds_and_failed_solves(ctx, tmp_path):
    n, q, nd = 48, 7, 5
    out = str(tmp_path) + "/svd/"
    obs, prior, settings = make(n, q)
    settings['rM'], settings['oversample'] = 4, 2
    obs.problem.fail_every = 3                      # every third forward solve raises
    gen = hf.DataGenerator(obs, prior, settings=settings)
    gen.generate(nd, derivatives=(1, 0), data_dir=out, compress=True, clean_up=False)
    assert gen.exceptions_count >= 2 and len(os.listdir(out + "skipped/")) == gen.exceptions_count
    for i in range(nd):
        for name in ("mq_data/m_sample_%d", "mq_data/q_sample_%d", "J_data/U_sample_%d", "J_data/sigma_sample_%d", "J_data/V_sample_%d"):
            assert os.path.exists(out + name % i + ".npy")
    mq, js = np.load(out + "mq_data.npz"), np.load(out + "Jsvd_data.npz")
    assert sorted(mq.files) == ["m_data", "q_data"] and sorted(js.files) == ["U_data", "V_data", "sigma_data"]
    assert js["U_data"].shape == (nd, q, 4) and js["sigma_data"].shape == (nd, 4) and js["V_data"].shape == (nd, n, 4)
    obs.problem.fail_every = 0
    for i, qi, Jd, _ in jacobians_at(obs, mq["m_data"]):
        np.testing.assert_allclose(mq["q_data"][i], qi, rtol=1e-12)
        U, s, V = js["U_data"][i], js["sigma_data"][i], js["V_data"][i]
        sv = np.linalg.svd(Jd, compute_uv=False)
        assert rel(U.T @ Jd @ V, np.diag(s)) < 1e-9 and rel(U.T @ U, np.eye(4)) < 1e-10 and rel(V.T @ V, np.eye(4)) < 1e-10
        assert np.all(s <= sv[:4] * (1 + 1e-12)) and s[0] > 0.999 * sv[0]           # 6 probe columns for 7 rows: close to exact
    # the archive alone is left after a clean-up
    gen.generate(2, derivatives=(1, 0), data_dir=str(tmp_path) + "/clean/", compress=True, clean_up=True)
    assert sorted(os.listdir(str(tmp_path) + "/clean/")) == ["Jsvd_data.npz", "mq_data.npz"]


@pytest.mark.parametrize("columns,route", [(3, "host"), (7, "device")])
def test_generate_in_an_output_basis(ctx, tmp_path, columns, route):
    """J^T (M Phi): fewer basis vectors than observable rows -> the adjoint columns of the matrix-free Jacobian are the answer;
    otherwise the rows are streamed to HBM and contracted there.  Same numbers either way."""
    n, q, nd = 48, 7, 4
    out = str(tmp_path) + "/"
    obs, prior, settings = make(n, q)
    rng = np.random.default_rng(1)
    Phi = np.linalg.qr(rng.standard_normal((q, columns)))[0]
    MPhi = Phi * np.linspace(1.0, 2.0, q)[:, None]
    gen = hf.DataGenerator(obs, prior, settings=settings)
    gen.generate(nd, derivatives=(1, 0), output_decoder=Phi, output_encoder=MPhi, data_dir=out, clean_up=True)
    f = np.load(out + "JstarPhi_data.npz")
    assert sorted(f.files) == ["JstarPhi_data", "MPhi", "Phi"] and f["JstarPhi_data"].shape == (nd, n, columns)
    np.testing.assert_array_equal(f["Phi"], Phi)
    inc_before = obs.n_inc_solve
    for i, _, Jd, _ in jacobians_at(obs, np.load(out + "mq_data.npz")["m_data"]):
        assert rel(f["JstarPhi_data"][i], Jd.T @ MPhi) < 1e-10
    assert obs.n_inc_solve == inc_before                   # (the check itself solves nothing incrementally)
    assert inc_before == nd * (columns if route == "host" else q)


@pytest.mark.parametrize("columns", [3, 10])
def test_generate_in_an_input_basis(ctx, tmp_path, columns):
    n, q, nd = 48, 7, 3
    out = str(tmp_path) + "/"
    obs, prior, settings = make(n, q)
    Psi = np.linalg.qr(np.random.default_rng(2).standard_normal((n, columns)))[0]
    gen = hf.DataGenerator(obs, prior, settings=settings)
    gen.generate(nd, derivatives=(1, 0), input_decoder=Psi, input_encoder=Psi, data_dir=out, clean_up=True)
    f = np.load(out + "JPsi_data.npz")
    assert sorted(f.files) == ["JPsi_data", "Psi", "input_encoder"] and f["JPsi_data"].shape == (nd, q, columns)
    for i, _, Jd, _ in jacobians_at(obs, np.load(out + "mq_data.npz")["m_data"]):
        assert rel(f["JPsi_data"][i], Jd @ Psi) < 1e-10
    assert obs.n_inc_solve == nd * min(columns, q)


def test_generate_for_a_control_problem(ctx, tmp_path):
    n, q, dz, nd = 40, 6, 4, 3
    out = str(tmp_path) + "/"
    obs, prior, settings = make(n, q, dz)
    settings['rM'], settings['rZ'], settings['oversample'] = 3, 2, 10
    gen = hf.DataGenerator(obs, prior, control_distribution=fp.ControlDistribution(dz), settings=settings)
    gen.generate(nd, derivatives=(1, 1), data_dir=out, clean_up=True)
    assert sorted(os.listdir(out)) == ["Jsvd_data.npz", "Jzsvd_data.npz", "mzq_data.npz"]
    mzq, jz, js = np.load(out + "mzq_data.npz"), np.load(out + "Jzsvd_data.npz"), np.load(out + "Jsvd_data.npz")
    assert sorted(mzq.files) == ["m_data", "q_data", "z_data"] and sorted(jz.files) == ["Uz_data", "Vz_data", "sigmaz_data"]
    assert jz["Uz_data"].shape == (nd, q, 2) and jz["Vz_data"].shape == (nd, dz, 2) and js["V_data"].shape == (nd, n, 3)
    for i, qi, Jd, Jzd in jacobians_at(obs, mzq["m_data"], mzq["z_data"]):
        np.testing.assert_allclose(mzq["q_data"][i], qi, rtol=1e-12)
        # four probe columns span the whole control space, six the whole observable space: exact truncated SVDs
        np.testing.assert_allclose(jz["sigmaz_data"][i], np.linalg.svd(Jzd, compute_uv=False)[:2], rtol=1e-9)
        np.testing.assert_allclose(js["sigma_data"][i], np.linalg.svd(Jd, compute_uv=False)[:3], rtol=1e-9)
        assert rel(jz["Uz_data"][i].T @ Jzd @ jz["Vz_data"][i], np.diag(jz["sigmaz_data"][i])) < 1e-9
    # both derivatives in an output basis
    Phi = np.linalg.qr(np.random.default_rng(3).standard_normal((q, 2)))[0]
    out2 = str(tmp_path) + "/basis/"
    gen.generate(nd, derivatives=(1, 1), output_decoder=Phi, data_dir=out2, clean_up=True)
    fz, f = np.load(out2 + "JzstarPhi_data.npz"), np.load(out2 + "JstarPhi_data.npz")
    assert sorted(fz.files) == ["JzstarPhi_data", "MPhi", "Phi"]
    mzq = np.load(out2 + "mzq_data.npz")
    for i, _, Jd, Jzd in jacobians_at(obs, mzq["m_data"], mzq["z_data"]):
        assert rel(fz["JzstarPhi_data"][i], Jzd.T @ Phi) < 1e-10 and rel(f["JstarPhi_data"][i], Jd.T @ Phi) < 1e-10


def test_generate_for_a_control_problem_survives_a_failing_save(ctx, tmp_path, monkeypatch):
    """An m / q / z save that raises AFTER the derivative work of the sample was queued: the sample is drawn again and the queued
    control Jacobian of the discarded attempt must not stay behind (one matrix per stored sample, rows aligned with their index)."""
    n, q, dz, nd = 40, 6, 4, 4
    out = str(tmp_path) + "/"
    obs, prior, settings = make(n, q, dz)
    settings['rM'], settings['rZ'], settings['oversample'] = 3, 2, 10
    gen = hf.DataGenerator(obs, prior, control_distribution=fp.ControlDistribution(dz), settings=settings)
    real_save, fired = np.save, []

    def flaky_save(path, values, *a, **k):
        if str(path).endswith("q_sample_1.npy") and not fired:
            fired.append(path)
            raise OSError("disk full (injected)")
        return real_save(path, values, *a, **k)

    monkeypatch.setattr(np, "save", flaky_save)
    Phi = np.linalg.qr(np.random.default_rng(3).standard_normal((q, 2)))[0]
    gen.generate(nd, derivatives=(1, 1), output_decoder=Phi, data_dir=out, clean_up=True)
    monkeypatch.undo()
    assert fired and gen.exceptions_count == 1
    fz, f, mzq = np.load(out + "JzstarPhi_data.npz"), np.load(out + "JstarPhi_data.npz"), np.load(out + "mzq_data.npz")
    assert fz["JzstarPhi_data"].shape[0] == nd and mzq["m_data"].shape[0] == nd
    for i, qi, Jd, Jzd in jacobians_at(obs, mzq["m_data"], mzq["z_data"]):
        np.testing.assert_allclose(mzq["q_data"][i], qi, rtol=1e-12)
        assert rel(fz["JzstarPhi_data"][i], Jzd.T @ Phi) < 1e-10 and rel(f["JstarPhi_data"][i], Jd.T @ Phi) < 1e-10


def test_two_step_generate_for_a_full_state_problem(ctx, tmp_path):
    """States first, their POD (in the mass inner product), then J^T (M phi) at the stored samples (dataGenerator.py:251-356)."""
    n, nd, r = 40, 8, 4
    out = str(tmp_path) + "/"
    h = 1.0 / (n + 1)
    M = sp.diags([np.full(n - 1, h / 6), np.full(n, 4 * h / 6), np.full(n - 1, h / 6)], [-1, 0, 1], format="csr")
    prob = fp.NumpyProblem(n, hf.HostVector)
    obs = fp.ProtocolObservable(prob, hf.StateSpaceIdentityOperator(fp.MatrixOperator(M)))
    prior = fp.NumpyPrior(n, hf.HostVector)
    hf.parRandom.reseed(5)
    hf.parRandom.split(0)
    settings = hf.data_generator_settings()
    settings['verbose'] = False
    gen = hf.DataGenerator(obs, prior, settings=settings)
    gen.two_step_generate(nd, derivatives=(1, 0), pod_rank=r, data_dir=out, clean_up=True)
    assert sorted(os.listdir(out)) == ["JstarPhi_data.npz", "POD", "mq_data.npz"]
    assert sorted(os.listdir(out + "POD")) == ["POD_decoder.npy", "POD_encoder.npy", "POD_shift.npy", "d_POD.npy"]
    phi, Mphi = np.load(out + "POD/POD_decoder.npy"), np.load(out + "POD/POD_encoder.npy")
    assert rel(Mphi, M @ phi) < 1e-12 and rel(phi[:, :r - 1].T @ Mphi[:, :r - 1], np.eye(r - 1)) < 1e-8
    mq, f = np.load(out + "mq_data.npz"), np.load(out + "JstarPhi_data.npz")
    np.testing.assert_allclose(np.load(out + "POD/POD_shift.npy"), mq["q_data"].mean(axis=0), rtol=1e-12)
    u, m = obs.generate_vector(hf.STATE), obs.generate_vector(hf.PARAMETER)
    for i in range(nd):
        m.set_local(mq["m_data"][i])
        obs.solveFwd(u, [u, m, None])
        np.testing.assert_allclose(mq["q_data"][i], u.get_local(), rtol=1e-12)
        obs.setLinearizationPoint([u, m, None])
        Jd = prob.jacobian_dense(np.eye(n))
        # the adjoint of the identity observation in the mass inner product is M (fullStateObservable.py:41-52)
        assert rel(f["JstarPhi_data"][i], Jd.T @ (M @ Mphi)) < 1e-9
    with pytest.raises(AssertionError):              # not a full-state problem
        obs2 = fp.ProtocolObservable(prob, fp.MatrixOperator(fp.observation_matrix(5, n)))
        hf.DataGenerator(obs2, prior, settings=settings).two_step_generate(4, pod_rank=2, data_dir=out + "x/")


def test_matrix_free_svd_route_for_a_full_state_observable(ctx, tmp_path):
    """q = n rows and a rank-2 factorisation with 2 extra probe columns: 16 incremental solves per sample through the matrix-free
    Jacobian instead of the 40 that materialising it would take; the device's randomized SVD drives the host operator."""
    n, nd = 40, 2
    out = str(tmp_path) + "/"
    prob = fp.NumpyProblem(n, hf.HostVector)
    obs = fp.ProtocolObservable(prob, hf.StateSpaceIdentityOperator(fp.MatrixOperator(sp.identity(n, format="csr")), use_mass_matrix=False))
    prior = fp.NumpyPrior(n, hf.HostVector)
    hf.parRandom.reseed(6)
    hf.parRandom.split(0)
    settings = hf.data_generator_settings()
    settings['verbose'], settings['rM'], settings['oversample'] = False, 2, 2
    hf.DataGenerator(obs, prior, settings=settings).generate(nd, derivatives=(1, 0), data_dir=out, clean_up=True)
    assert obs.n_inc_solve == nd * 16
    js, mq = np.load(out + "Jsvd_data.npz"), np.load(out + "mq_data.npz")
    assert js["U_data"].shape == (nd, n, 2) and js["V_data"].shape == (nd, n, 2)
    u, m = obs.generate_vector(hf.STATE), obs.generate_vector(hf.PARAMETER)
    for i in range(nd):
        m.set_local(mq["m_data"][i])
        obs.solveFwd(u, [u, m, None])
        obs.setLinearizationPoint([u, m, None])
        Jd = prob.jacobian_dense(np.eye(n))
        U, s, V = js["U_data"][i], js["sigma_data"][i], js["V_data"][i]
        assert rel(U.T @ Jd @ V, np.diag(s)) < 1e-9 and rel(U.T @ U, np.eye(2)) < 1e-10 and rel(V.T @ V, np.eye(2)) < 1e-10
        assert s[0] > 0.9 * np.linalg.svd(Jd, compute_uv=False)[0]
