#!/usr/bin/env python3
"""Headline benchmark: randomized double-pass eigensolve throughput (GDoF*rank/s) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload as|pod|kle] [--quick]

One "step" = one full double pass (Omega already in HBM -> eigenvalues on the host, eigenvectors in
HBM) over one synthetic workload whose operator data is resident in HBM.  The default workload is
BASELINE config 4 (ActiveSubspaceProjector: 512 Monte-Carlo Jacobian samples of 100 x 2e5, r=64, p=10):
it is the configuration the metric's "1/2/4/8 GPU" clause is quoted on, it fits one GPU at N=1 (82 GB of
Jacobians in 288 GB of HBM) and it is the one path with a real exchange step, so the SAME total work is
timed at every N (strong scaling): the samples are sharded 512/N per rank and the block J^T J Omega is
all-reduced (RCCL over xGMI) once per operator application.

N > 1 runs one rank per GPU over the native communicator of libhfmi (RCCL over xGMI; no torch in this file).  Either
launch works: plain ``python bench.py --gpus N`` (this process then only spawns the N ranks and touches no GPU), or
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` (RANK / LOCAL_RANK / WORLD_SIZE from
the environment).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X datasheet, dense FP64 matrix (= FP64 vector) peak
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="as", choices=["as", "pod", "kle"])
    ap.add_argument("--quick", action="store_true", help="1/8-size problem (smoke / profiling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--samples-total", type=int, default=512,
                    help="config 4 only: total Monte-Carlo samples (512 = BASELINE; 64 on one GPU reproduces the per-GPU "
                         "share of the 8-GPU run, for estimating the non-scaling part)")
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "p2p"],
                    help="device transport of the native communicator (HFMI_COMM_TRANSPORT): auto = RCCL over xGMI when every "
                         "rank has its own GPU, direct peer access (HIP IPC) when ranks SHARE a GPU -- the functional check of "
                         "the sharded path on a 1-GPU box")
    ap.add_argument("--dist-single", action="store_true",
                    help="exercise the multi-GPU code path (native communicator, all-reduce inside the fused solve) with one rank")
    return ap.parse_args()


def build_workload(args, hf, rank, world):
    from hippyflow_amd import workloads
    scale = 8 if args.quick else 1
    if args.workload == "as":
        N, ns_total, q, r, p = 200000 // scale, args.samples_total, 100, 64, 10
        assert ns_total % world == 0
        ns_local = ns_total // world
        wl = workloads.as_workload(N, ns_local, q=q, latent=q, rate=0.06, seed=4, first_sample=rank * ns_local, ns_total=ns_total)
        desc = {"workload": "config4 ActiveSubspaceProjector: mean J^T J, %d samples x (%d x %d), r=%d, p=%d" % (ns_total, q, N, r, p),
                "N": N, "samples_total": ns_total, "samples_per_gpu": ns_local, "outputs": q, "rank": r, "oversampling": p,
                "parallelism": "sample-parallel x%d, one all-reduce(avg) of the N x k block per operator application" % world}
        op = wl.operator
        B = Binv = None
    elif args.workload == "pod":
        N, n, r, p = 500000 // scale, 2048, 128, 10
        assert n % world == 0
        wl = workloads.pod_workload(N, n, latent=256, rate=0.05, seed=3, first_snapshot=rank * (n // world), n_local=n // world)
        desc = {"workload": "config3 PODProjector: %d snapshots x N=%d, r=%d, p=%d" % (n, N, r, p), "N": N, "snapshots": n,
                "snapshots_per_gpu": n // world, "rank": r, "oversampling": p,
                "parallelism": "single GPU" if world == 1 else "snapshot-parallel x%d, one all-reduce(avg) of the N x k block per operator application" % world}
        op = wl.operator
        B = Binv = None
    else:
        nx, ny, r, p = 400, 250 // scale, 64, 20
        wl = workloads.kle_workload(nx, ny, latent=256, rate=0.08, seed=2)
        N = wl.N
        if world > 1:
            raise SystemExit("kle workload: replicas only (KLEProjector.py:148-149); run at --gpus 1")
        desc = {"workload": "config2 KLEProjector(mass): dense covariance N=%d, r=%d, p=%d" % (N, r, p), "N": N, "rank": r,
                "oversampling": p, "parallelism": "single GPU"}
        op = hf.MassPreconditionedCovarianceOperator(wl.C_operator, wl.M_operator)
        B = wl.M_operator
        Binv = hf.CsrPCGSolver(wl.M_operator.csr)
    return wl, op, B, Binv, N, r, p, desc


def host_reference(args, wl, Omega_host, r, hf_o, hp_o):
    """The oracle on the SAME inputs, evaluated in factored form on the host (see workloads.py)."""
    from hippyflow_amd import workloads
    if args.workload == "as":
        P = wl.P.to_dense()
        H = workloads.as_reduced_matrix(4, wl.ns_total, wl.q, wl.latent, 0.06)   # all samples of all ranks
        apply_A = lambda W: np.asfortranarray(P @ (H @ (P.T @ W)))
        return hp_o.double_pass_blas3(apply_A, Omega_host, r)
    if args.workload == "pod":
        apply_A = workloads.pod_host_apply(wl, wl.W0.to_dense())
        return hp_o.double_pass_blas3(apply_A, Omega_host, r)
    import scipy.sparse.linalg as spla
    F, lam, M = wl.F_host, wl.lam, wl.M
    lu = spla.splu(M.tocsc())
    apply_A = lambda W: np.asfortranarray(M @ (F @ (lam[:, None] * (F.T @ (M @ W)))))
    return hp_o.double_pass_blas3(apply_A, Omega_host, r, apply_B=lambda W: M @ W, apply_Binv=lambda W: np.asfortranarray(lu.solve(np.ascontiguousarray(W))))


def cpu_baseline(args, wl, Omega_host, r, N, hp_o, hf_o):
    """CPU port of the reference path (BLAS-3 "best-effort" form) on a bounded sample."""
    cores = os.cpu_count() or 1
    k = Omega_host.shape[1]
    if args.workload == "as":
        ns_s = 16
        Jh = wl.J.view(0, ns_s * wl.q).to_vectors().reshape(ns_s, wl.q, N)     # dense Jacobians of the sample
        W = Omega_host
        t0 = time.perf_counter()
        Y = hf_o.mean_jtj_block_blas3(Jh, W)
        t_apply = time.perf_counter() - t0
        t0 = time.perf_counter()
        Q, _ = hp_o._qr_posdiag(Y)
        T = Y.T @ Q
        np.linalg.eigh(0.5 * (T + T.T))
        U = Q @ T[:, :r]
        t_rest = time.perf_counter() - t0
        t_full = 2.0 * t_apply * (wl.ns_total / ns_s) + t_rest
        sample = "dense BLAS-3 mean-JtJ apply timed on %d of %d samples (full N=%d, k=%d) and scaled linearly in samples, + QR/Rayleigh-Ritz at full size" % (ns_s, wl.ns_total, N, k)
    elif args.workload == "pod":
        n_s = 256
        Xh = wl.X.view(0, n_s).to_vectors()
        t0 = time.perf_counter()
        Y = hf_o.snapshot_gram_block(Xh, Omega_host)
        t_apply = time.perf_counter() - t0
        t0 = time.perf_counter()
        Q, _ = hp_o._qr_posdiag(Y)
        T = Y.T @ Q
        np.linalg.eigh(0.5 * (T + T.T))
        t_rest = time.perf_counter() - t0
        t_full = 2.0 * t_apply * (wl.n / n_s) + t_rest
        sample = "BLAS-3 snapshot-Gram apply timed on %d of %d snapshots and scaled linearly, + QR/Rayleigh-Ritz at full size" % (n_s, wl.n)
    else:
        rows = 4000
        Ch = wl.C.view(0, rows).to_vectors()
        t0 = time.perf_counter()
        _ = Ch @ Omega_host
        t_apply = time.perf_counter() - t0
        t_full = 2.0 * t_apply * (N / rows)
        sample = "dense C*Omega timed on %d of %d rows and scaled linearly (sparse M, M^-1 and QR not included)" % (rows, N)
    return {"value": N * r / t_full / 1e9, "unit": "GDoF*rank/s", "cores": cores, "kind": "port", "sample": sample,
            "seconds_full_estimate": t_full}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this parent touches no GPU; it starts N fresh child interpreters (one per
        # device, nothing is re-exec'ed) and rank 0's JSON line arrives on the inherited stdout
        from hippyflow_amd.launch import spawn_ranks
        sys.exit(spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    # stdout carries exactly ONE JSON line: anything libraries print (e.g. the RCCL version banner) goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if args.transport != "auto":
        os.environ["HFMI_COMM_TRANSPORT"] = args.transport

    import hippyflow_amd as hf
    if hf.device_count() < 1:
        raise SystemExit("bench.py needs a GPU (libhfmi has no CPU path)")
    ctx = hf.Context.default()                       # cuda:LOCAL_RANK
    use_dist = world > 1 or args.dist_single
    if world > 1:
        collective = hf.NativeCollective.from_env(ctx)          # id through $HFMI_COMM_ID_FILE / the launcher's temp file
    elif args.dist_single:
        collective = hf.NativeCollective.from_unique_id(hf.NativeCollective.unique_id(), 1, 0, ctx=ctx)
    else:
        collective = hf.NullCollective()

    wl, op, B, Binv, N, r, p, desc = build_workload(args, hf, rank, world)
    k = r + p
    A = hf.CollectiveOperator(op, collective, mpi_op="avg") if use_dist else op
    hf.parRandom.reseed(1)
    Omega = hf.MultiVector(N, k)
    hf.parRandom.normal(1.0, Omega)          # identical on every rank (counter-based RNG): no broadcast

    def step():
        if B is None:
            return hf.doublePass(A, Omega, r, s=1)
        return hf.doublePassG(A, B, Binv, Omega, r, s=1)

    def barrier():
        collective.barrier()                         # drains this rank's stream, then meets the other ranks
        ctx.synchronize()

    for _ in range(args.warmup):
        d, U = step()
    barrier()
    ctx.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        d, U = step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_end()
    if use_dist:
        elapsed = collective.allReduceMax(elapsed)   # the slowest rank's clock
    comm_info = {"ranks": collective.size(), "transport": getattr(collective, "transport", "none"),
                 "launcher": os.environ.get("HFMI_LAUNCHER", "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "none")}
    if rank != 0:
        collective.barrier()
        if use_dist:
            collective.close()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = N * r / (elapsed / args.steps) / 1e9
    out = {"metric": "randomized-SVD throughput (GDoF*rank/s)", "value": value, "unit": "GDoF*rank/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic (seeded latent-factor model, generated in HBM)",
           "config": desc, "communicator": comm_info}

    # roofline of the dominant kernel (= the (kernel, shape) group with the largest total time), from per-launch
    # HIP events recorded inside the timed region on the stream the kernels run on
    prof = [g for g in prof if g["launches"] > 0 and g["ms"] > 0]
    if prof:
        pk = max(prof, key=lambda g: g["ms"])
        avg_ms = pk["ms"] / pk["launches"]
        tflops = pk["flops_per_launch"] / (avg_ms * 1e-3) / 1e12
        gbs = pk["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
        ai = pk["flops_per_launch"] / pk["bytes_per_launch"]
        ridge = FP64_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        if ai >= ridge:
            out["roofline"] = {"bound": "mfma", "achieved": tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": tflops / FP64_MFMA_PEAK_TFLOPS, "traffic": None}
        else:
            out["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                               "traffic": None}
        out["roofline"].update({"kernel": "%s m=%d k=%d N=%d" % (pk["kernel"], pk["m"], pk["k"], pk["N"]), "avg_launch_ms": avg_ms,
                                "launches_per_step": pk["launches"] / args.steps,
                                "algorithmic_flops_per_launch": pk["flops_per_launch"],
                                "algorithmic_bytes_per_launch": pk["bytes_per_launch"],
                                "arithmetic_intensity_flop_per_byte": ai, "ridge_flop_per_byte": ridge,
                                "algorithmic_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                                "fp64_mfma_frac": tflops / FP64_MFMA_PEAK_TFLOPS})
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE x2 gfx950
        # correction + WRITE_SIZE; profiles/pmc_traffic.json) -- bench.py cannot run the profiler on itself
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            rec = tr["kernels"].get(out["roofline"]["kernel"])
            if rec:
                out["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = tr["source"]
                out["roofline"]["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / pk["bytes_per_launch"]
                out["roofline"]["pmc_mfma_pipe_util"] = rec.get("mfma_pipe_util")
                out["roofline"]["pmc_effective_clock_ghz"] = rec.get("effective_clock_ghz")
        except (OSError, ValueError, KeyError):
            pass
        mfma_ms = sum(g["ms"] for g in prof) / args.steps
        out["kernels"] = [{"kernel": g["kernel"], "m": g["m"], "k": g["k"], "N": g["N"], "ms_per_step": g["ms"] / args.steps,
                           "launches_per_step": g["launches"] / args.steps, "avg_launch_ms": g["ms"] / g["launches"],
                           "tflops": g["flops_per_launch"] * g["launches"] / (g["ms"] * 1e-3) / 1e12}
                          for g in sorted(prof, key=lambda g: -g["ms"])]
        out["ms_per_step_in_mfma_kernels"] = mfma_ms
    try:
        out["device_peaks_measured"] = ctx.bench_peaks()
    except Exception as exc:   # the micro-benchmark is informative only
        out["device_peaks_measured"] = {"error": str(exc)}

    from oracle import hippyflow_restated as hf_o
    from oracle import hippylib_restated as hp_o
    Omega_host = np.asfortranarray(Omega.to_dense())
    if not args.no_check:
        t0 = time.perf_counter()
        d_ref, U_ref = host_reference(args, wl, Omega_host, r, hf_o, hp_o)
        Ud = np.asfortranarray(U.to_dense())
        lead = max(1, r // 2)
        if B is None:
            angle = hp_o.principal_angle(Ud[:, :lead], U_ref[:, :lead])
        else:
            angle = hp_o.principal_angle(Ud[:, :lead], U_ref[:, :lead], lambda W: wl.M @ W)
        out["parity"] = {"eig_rel_err_vs_oracle": hp_o.eig_rel_err(d, d_ref), "principal_angle_rad_leading_%d" % lead: angle,
                         "eigenvalue_range": [float(d[0]), float(d[-1])], "oracle_seconds": time.perf_counter() - t0,
                         "note": "oracle = CPU restatement of the reference path on the same Omega and the same operator (factored form)"}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, wl, Omega_host, r, N, hp_o, hf_o)
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)                      # whatever the libraries print while shutting down stays off stdout
    collective.barrier()
    if use_dist:
        collective.close()


if __name__ == "__main__":
    main()
