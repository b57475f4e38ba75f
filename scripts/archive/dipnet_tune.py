"""Tuning pass for the config-5 surrogate: device projectors once, then GPU trainings over (epochs, lr, schedule, ranks).
    python scripts/dipnet_tune.py"""
import sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, ".")
from hippyflow_amd import workloads, ActiveSubspaceParameterList, ActiveSubspaceProjector, PODParameterList, PODProjector
from hippyflow_amd.io_utils import get_projectors, modify_projectors
from hippyflow_amd.surrogate import ProjectedLowRankResidualNetwork, relative_l2_error
from torch import nn

wl = workloads.dipnet_workload(dM=20000, dQ=400, hidden=80, n_train=8192, n_test=1024, ns=64)
out = tempfile.mkdtemp() + "/"
ap = ActiveSubspaceParameterList(); ap["rank"], ap["oversampling"], ap["samples_per_process"] = 50, 10, wl.ns
ap["serialized_sampling"], ap["verbose"], ap["output_directory"] = False, False, out
ActiveSubspaceProjector(wl.observable, None, parameters=ap).construct_input_subspace(prior_preconditioned=False)
pp = PODParameterList(); pp["rank"], pp["oversampling"], pp["verbose"], pp["output_directory"] = 50, 10, False, out
pod = PODProjector(wl.observable, None, parameters=pp); pod.set_snapshots(wl.q_train[:2048].astype(np.float64)); pod.construct_subspace()
Vin, Uout = modify_projectors(get_projectors(out, fixed_input_rank=50, fixed_output_rank=50), 'as', 'pod')
dev = torch.device("cuda", 0)
mt, qt = torch.from_numpy(wl.m_train).to(dev), torch.from_numpy(wl.q_train).to(dev)
ms, qs = torch.from_numpy(wl.m_test).to(dev), torch.from_numpy(wl.q_test).to(dev)
# floor: best possible with the two projections (oracle reduced map unknown) -> output projection error alone
Uo = torch.from_numpy(np.ascontiguousarray(Uout)).float().to(dev)
proj = (qs @ Uo) @ torch.linalg.pinv(Uo).T if False else qs @ Uo @ torch.linalg.pinv(Uo)
Vi = torch.from_numpy(np.ascontiguousarray(Vin)).float().to(dev)
W1 = torch.from_numpy(wl.W1).to(dev); W2 = torch.from_numpy(wl.W2).to(dev)
Vo = torch.linalg.qr(Vi)[0]
mp = (ms @ Vo) @ Vo.T
qp = torch.tanh(mp @ W1) @ W2.T
print("orthonormality of the input projector: %.2e" % float(torch.linalg.norm(Vi.T @ Vi - torch.eye(Vi.shape[1], device=dev))))
print("input projection floor (true map at V V^T m): %.4f" % float((torch.linalg.norm(qs - qp, dim=1) / torch.linalg.norm(qs, dim=1)).mean()))
Pd = torch.from_numpy(wl.P.to_dense()).float().to(dev)
print("part of span(P) captured by V: singular values of P^T V: min %.3f, number > 0.99: %d" % (float(torch.linalg.svdvals(Pd.T @ Vi).min()), int((torch.linalg.svdvals(Pd.T @ Vi) > 0.99).sum())))
print("output projection floor: %.4f" % float((torch.linalg.norm(qs - proj, dim=1) / torch.linalg.norm(qs, dim=1)).mean()))

def train(epochs, lr, sched, ranks, bs=256, wd=0.0):
    torch.manual_seed(0)
    net = ProjectedLowRankResidualNetwork(Vin, Uout, ranks=list(ranks)).to(dev)
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=lr, weight_decay=wd)
    n = mt.shape[0]; steps = epochs * ((n + bs - 1) // bs)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=lr, total_steps=steps) if sched == "onecycle" else (
        torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=steps) if sched == "cos" else None)
    gen = torch.Generator(device="cpu").manual_seed(0)
    t0 = time.perf_counter()
    for ep in range(epochs):
        perm = torch.randperm(n, generator=gen).to(dev)
        for i in range(0, n, bs):
            idx = perm[i:i + bs]
            opt.zero_grad(set_to_none=True)
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                pred = net(mt[idx])
            loss = nn.functional.mse_loss(pred.float(), qt[idx])
            loss.backward(); opt.step()
            if sch: sch.step()
    torch.cuda.synchronize()
    return relative_l2_error(net, ms, qs), relative_l2_error(net, mt[:1024], qt[:1024]), time.perf_counter() - t0

zs = (mt[:2048] @ Vi)
scale = float(zs.pow(2).mean().sqrt())
print("rms of the projected training inputs: %.4f -> input projector divided by it" % scale)
Vin = Vin / scale
for cfg in [(100, 2e-3, "cos", (32, 32)), (300, 2e-3, "cos", (64, 64)), (300, 2e-3, "cos", (32, 32, 32, 32)), (300, 2e-3, "cos", (50, 50, 50)), (600, 2e-3, "cos", (64, 64, 64))]:
    te, tr, t = train(*cfg)
    print("epochs %4d lr %.0e sched %-8s ranks %-18s test %.4f train %.4f  %.1f s" % (cfg[0], cfg[1], cfg[2], cfg[3], te, tr, t), flush=True)
