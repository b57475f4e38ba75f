"""Config 5: the projected-network surrogate.  CPU tests cover the architecture contract
(frozen input projection initialised with the decoder, output layer initialised with the POD basis); the GPU test
trains under bf16 autocast and compares with an fp32 CPU evaluation of the same PyTorch model (parity with the
keras reference is unpinned: TensorFlow cannot run here)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _problem(seed=0, dM=400, dQ=120, r_in=12, r_out=10, n=1536):
    rng = np.random.default_rng(seed)
    Vin, _ = np.linalg.qr(rng.standard_normal((dM, r_in)))
    Uout, _ = np.linalg.qr(rng.standard_normal((dQ, r_out)))
    W = rng.standard_normal((r_in, r_out)) / np.sqrt(r_in)
    m = rng.standard_normal((n, dM))
    z = m @ Vin
    q = (np.tanh(z @ W) + 0.1 * (z @ W)) @ Uout.T
    return Vin, Uout, m.astype(np.float32), q.astype(np.float32)


def test_architecture_contract():
    from hippyflow_amd.surrogate import ProjectedDense, ProjectedLowRankResidualNetwork
    Vin, Uout, m, q = _problem()
    net = ProjectedLowRankResidualNetwork(Vin, Uout, ranks=[4, 4])
    np.testing.assert_allclose(net.input_proj_layer.weight.detach().numpy(), Vin.T.astype(np.float32), rtol=1e-6)
    np.testing.assert_allclose(net.output_layer.weight.detach().numpy(), Uout.astype(np.float32), rtol=1e-6)
    assert not net.input_proj_layer.weight.requires_grad and net.output_layer.weight.requires_grad
    assert float(net.output_layer.bias.abs().sum()) == 0.0 and float(net.input_bias.bias.abs().sum()) == 0.0
    x = torch.from_numpy(m[:5])
    assert net(x).shape == (5, Uout.shape[0])
    assert ProjectedDense(Vin, Uout, intermediate_layers=2)(x).shape == (5, Uout.shape[0])
    # with zeroed residual blocks and an identity-like reduction the network is U_out R V_in^T: the projected linear map
    with torch.no_grad():
        for blk in net.blocks:
            blk.up.weight.zero_()
            blk.up.bias.zero_()
        net.reduced.weight.copy_(torch.eye(Uout.shape[1], Vin.shape[1]))
        net.reduced.bias.zero_()
    np.testing.assert_allclose(net(x).detach().numpy(), (m[:5] @ Vin)[:, :Uout.shape[1]] @ Uout.T, rtol=2e-4, atol=2e-5)


def test_training_reduces_error_cpu():
    from hippyflow_amd.surrogate import ProjectedLowRankResidualNetwork, l2_accuracy, train_surrogate
    torch.manual_seed(0)
    Vin, Uout, m, q = _problem(n=512)
    mt, qt = torch.from_numpy(m), torch.from_numpy(q)
    net = ProjectedLowRankResidualNetwork(Vin, Uout, ranks=[8, 8])
    a0 = l2_accuracy(net, mt, qt)
    hist = train_surrogate(net, mt, qt, epochs=15, batch_size=64, lr=3e-3)
    assert hist[-1] < 0.5 * hist[0] and l2_accuracy(net, mt, qt) > a0


@pytest.mark.gpu
def test_bf16_training_on_gpu_matches_fp32_evaluation():
    from hippyflow_amd.surrogate import ProjectedLowRankResidualNetwork, l2_accuracy, train_surrogate
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible to torch")
    torch.manual_seed(0)
    Vin, Uout, m, q = _problem(dM=2000, dQ=300, r_in=50, r_out=50, n=4096)
    dev = torch.device("cuda", 0)
    mt, qt = torch.from_numpy(m).to(dev), torch.from_numpy(q).to(dev)
    net = ProjectedLowRankResidualNetwork(Vin, Uout, ranks=[16, 16]).to(dev)
    acc0 = l2_accuracy(net, mt[3584:], qt[3584:])
    hist = train_surrogate(net, mt[:3584], qt[:3584], epochs=60, batch_size=128, lr=3e-3, bf16=True)
    acc_gpu = l2_accuracy(net, mt[3584:], qt[3584:])
    assert hist[-1] < 0.2 * hist[0] and acc_gpu > max(0.7, acc0 + 0.3), (acc0, acc_gpu)
    # the same trained weights evaluated in fp32 on the CPU: the bf16 forward agrees to bf16 round-off
    cpu = ProjectedLowRankResidualNetwork(Vin, Uout, ranks=[16, 16])
    cpu.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        y_bf16 = net(mt[3584:3600]).float().cpu()
    y_fp32 = cpu(torch.from_numpy(m[3584:3600])).detach()
    assert float(torch.linalg.norm(y_bf16 - y_fp32) / torch.linalg.norm(y_fp32)) < 3e-2
    assert abs(l2_accuracy(cpu, torch.from_numpy(m[3584:]), torch.from_numpy(q[3584:])) - acc_gpu) < 1e-3


@pytest.mark.gpu
def test_config5_end_to_end_device_projectors_feed_the_network(tmp_path):
    """BASELINE config 5: device AS(50) and POD(50) solves -> the reference's .npy files -> get_projectors /
    modify_projectors -> ProjectedLowRankResidualNetwork, bf16 on the GPU vs the same restatement in fp32 on the CPU."""
    from hippyflow_amd import workloads
    from hippyflow_amd.surrogate import run_config5
    wl = workloads.dipnet_workload(dM=6000, dQ=200, hidden=80, n_train=4096, n_test=512, ns=32)
    res = run_config5(wl, str(tmp_path), r_in=50, r_out=50, epochs=250, batch_size=128, lr=2e-3, ranks=(64, 64, 64), cpu_epochs=6)
    assert res["input_projector_shape"] == [6000, 50] and res["output_projector_shape"] == [200, 50]
    assert res["AS_eigenvalues_first_last"][0] > res["AS_eigenvalues_first_last"][1] > 0
    raw, floors = res["untrained_rel_l2_test_error"], res["projection_floors"]
    gpu = res["gpu_bf16_rel_l2_test_error"]
    gpu_pre, cpu_pre = res["gpu_bf16_rel_l2_test_error_after_cpu_epochs"], res["cpu_fp32_rel_l2_test_error_after_cpu_epochs"]
    assert floors["input_projection_rel_l2"] < 0.1 and floors["output_projection_rel_l2"] < 0.1, res   # the device projectors carry the map
    assert gpu < 0.2 * raw and gpu < 0.2, res                             # the network learns it (round 2: 0.41 of 1.00)
    assert gpu_pre < 0.5 * raw and cpu_pre < 0.5 * raw, res
    assert abs(gpu_pre - cpu_pre) < 0.03, res                             # bf16 autocast tracks the fp32 run over the same epochs
    assert res["gpu_samples_per_second"] > 0
