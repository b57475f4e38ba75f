#!/usr/bin/env python3
"""Headline benchmark: randomized double-pass eigensolve throughput (GDoF*rank/s) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload as|pod|kle] [--prior] [--quick]

One "step" = one full double pass (Omega already in HBM -> eigenvalues on the host, eigenvectors in
HBM) over one synthetic workload whose operator data is resident in HBM.  The default workload is
BASELINE config 4 (ActiveSubspaceProjector: 512 Monte-Carlo Jacobian samples of 100 x 2e5, r=64, p=10,
J_i = A_i P^T + 0.01 E_i as SURVEY.md section 8d specifies): it is the configuration the metric's "1/2/4/8 GPU"
clause is quoted on, it fits one GPU at N=1 (82 GB of Jacobians in 288 GB of HBM) and it is the one path with a
real exchange step, so the SAME total work is timed at every N (strong scaling): the samples are sharded 512/N
per rank and the block J^T J Omega is all-reduced (RCCL over xGMI) once per operator application.
``--prior`` runs the reference's DEFAULT form of that solve (construct_input_subspace(prior_preconditioned=True),
activeSubspaceProjector.py:447-453): doublePassG with B = prior.R (CSR on the device) and B^-1 = prior.Rsolver,
a HOST sparse-LU black box reached through the pinned, pipelined callback path.

N > 1 runs one rank per GPU over the native communicator of libhfmi (RCCL over xGMI; no torch in this file).  Either
launch works: plain ``python bench.py --gpus N`` (this process then only spawns the N ranks and touches no GPU), or
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` (RANK / LOCAL_RANK / WORLD_SIZE from
the environment).  Rank 0 prints ONE JSON line.

Parity (``parity`` in the JSON): the CPU oracle (oracle/) runs the same algorithm on the SAME inputs -- the operator
data is streamed from HBM to the host in slabs and applied densely there (no knowledge of how it was generated).
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
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="as", choices=["as", "pod", "kle", "dipnet"],
                    help="as / pod / kle: BASELINE configs 4 / 3 / 2 of the projector path; dipnet: config 5, the projected-network "
                         "surrogate (PyTorch-ROCm bf16) fed by device AS x POD solves -- a separate JSON line with its own metric")
    ap.add_argument("--prior", action="store_true",
                    help="config 4, prior-preconditioned (the reference's default): doublePassG with B = R = A M_l^-1 A as CSR "
                         "on the device and B^-1 = a host sparse-LU callback")
    ap.add_argument("--noise", type=float, default=0.01, help="config 4: J_i = A_i P^T + noise * E_i (SURVEY 8d: 0.01; 0 = exactly rank 100)")
    ap.add_argument("--quick", action="store_true", help="1/8-size problem (smoke / profiling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-workers", type=int, default=max(1, min(16, (os.cpu_count() or 1) // 4)),
                    help="--prior: worker PROCESSES (each with its own sparse LU) the host R^-1 callback deals its vectors to")
    ap.add_argument("--ingest", action="store_true", help="report host -> HBM ingest rates instead of the solve (never part of `value`)")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-literal", action="store_true", help="skip the extra literal-T steps after the timed region")
    ap.add_argument("--samples-total", type=int, default=512,
                    help="config 4 only: total Monte-Carlo samples (512 = BASELINE; 64 on one GPU reproduces the per-GPU "
                         "share of the 8-GPU run, for estimating the non-scaling part)")
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "p2p"],
                    help="device transport of the native communicator (HFMI_COMM_TRANSPORT): auto = RCCL over xGMI when every "
                         "rank has its own GPU, direct peer access (HIP IPC) when ranks SHARE a GPU -- the functional check of "
                         "the sharded path on a 1-GPU box")
    ap.add_argument("--dist-single", action="store_true",
                    help="exercise the multi-GPU code path (native communicator, all-reduce inside the fused solve) with one rank")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed headline workload: skip the extra evidence lines (kernel point, configs 3 and 2, the 64-sample "
                         "shard step) that a default --gpus 1 run of config 4 appends as extra keys (A/B runs, profiling)")
    ap.add_argument("--kernel-point", action="store_true",
                    help="north-star kernel point instead of a solve: G = X^T Omega at N = 1e6, k = 138, n in {8 ... 2048}; one JSON line")
    ap.add_argument("--eig-large", action="store_true",
                    help="the whole-GPU symmetric eigensolver behind the deterministic POD (la.eigh(G), PODProjector.py:812-833) instead of a "
                         "solve: n = 512 ... 8192, wall time of hfmi_sym_eig_small next to numpy.linalg.eigh on the host; one JSON line")
    ap.add_argument("--cpu-baseline", default="full", choices=["full", "quick"],
                    help="quick: the host legs at all threads and at one socket's cores only (no single-thread legs)")
    return ap.parse_args()


AS_RATE, AS_SEED = 0.06, 4


def build_workload(args, hf, rank, world):
    from hippyflow_amd import workloads
    scale = 8 if args.quick else 1
    prior = None
    if args.workload == "as":
        nx, ny = 500, 400 // scale
        N, ns_total, q, r, p = nx * ny, args.samples_total, 100, 64, 10
        assert ns_total % world == 0
        ns_local = ns_total // world
        wl = workloads.as_workload(N, ns_local, q=q, latent=q, rate=AS_RATE, seed=AS_SEED, first_sample=rank * ns_local,
                                   ns_total=ns_total, noise=args.noise)
        desc = {"workload": "config4 ActiveSubspaceProjector%s: mean J^T J, %d samples x (%d x %d), J_i = A_i P^T + %g E_i, r=%d, p=%d"
                            % (" (prior-preconditioned, doublePassG)" if args.prior else "", ns_total, q, N, args.noise, r, p),
                "N": N, "samples_total": ns_total, "samples_per_gpu": ns_local, "outputs": q, "rank": r, "oversampling": p,
                "noise": args.noise,
                "parallelism": "sample-parallel x%d, one all-reduce(avg) of the N x k block per operator application" % world}
        op = wl.operator
        B = Binv = None
        if args.prior:
            prior = workloads.BiLaplacianPrior(nx, ny, delta=1.0, gamma=0.1, processes=args.host_workers)
            B = hf.CsrOperator(prior.R)
            # every worker process should get at least one vector of a slab: the whole block in one slab when there are many
            Binv = hf.HostCallbackOperator(prior.Rsolver, N, chunk_vectors=(0 if args.host_workers > 8 else None))
            desc.update({"B": "R = A M_l^-1 A, A = M + 0.1 K on a %d x %d P1 grid: CSR on the device, %.1f nnz/row" % (nx, ny, prior.R.nnz / N),
                         "Binv": "host sparse LU of A (SuperLU), two triangular sweeps per vector in %d worker process(es) (each with its own "
                                 "factorisation, slabs through shared memory), slabs of %s vectors through pinned double buffers"
                                 % (prior.Rsolver.processes, Binv.chunk_vectors or "all")})
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
        if world > 1:
            raise SystemExit("kle workload: replicas only (KLEProjector.py:148-149); run at --gpus 1")
        nx, ny, N, r, p = 316, 317, 100000 // (scale * scale), 64, 20
        wl = workloads.kle_matern_workload(nx, ny, N=N, sigma=1.0, ell=0.1)
        desc = {"workload": "config2 KLEProjector(mass): dense Matern-3/2 covariance (sigma=1, ell=0.1) on N=%d nodes of a %d x %d grid, r=%d, p=%d"
                            % (N, nx, ny, r, p), "N": N, "rank": r, "oversampling": p, "parallelism": "single GPU"}
        op = hf.MassPreconditionedCovarianceOperator(wl.C_operator, wl.M_operator)
        B = wl.M_operator
        Binv = hf.CsrPCGSolver(wl.M_operator.csr)
    return wl, op, B, Binv, prior, N, r, p, desc


# ---------------------------------------------------------------------------------------------- host side (oracle legs)
def _stream_rows(block, rows_per):
    """Slabs (first, count, host array (count, N)) of a device block, one vector per row."""
    n = block.nvec()
    for first in range(0, n, rows_per):
        cnt = min(rows_per, n - first)
        yield first, cnt, block.view(first, cnt).to_vectors()


def dense_streaming_operator(args, hf, wl, world):
    """apply_A(W) on the host from the operator DATA: slabs of the device-resident Jacobians / snapshots / covariance
    rows are copied to the host and contracted there (BLAS-3).  With several ranks, rank 0 regenerates the other ranks'
    shards slab by slab on its own GPU (the generator is keyed by the global sample index)."""
    from hippyflow_amd import workloads
    if args.workload == "as":
        q, ns_total = wl.q, wl.ns_total
        per = 32                                            # samples per slab: 32 x 100 x 2e5 doubles = 5.1 GB

        def slabs():
            if world == 1:
                for _, cnt, Jh in _stream_rows(wl.J, per * q):
                    yield Jh
            else:
                for first in range(0, ns_total, per):
                    cnt = min(per, ns_total - first)
                    part = workloads.as_workload(wl.N, cnt, q=q, latent=wl.latent, rate=AS_RATE, seed=AS_SEED, first_sample=first,
                                                 ns_total=ns_total, noise=wl.noise, P=wl.P)
                    yield part.J.to_vectors()

        def apply_A(W):
            Y = np.zeros((wl.N, W.shape[1]))
            for Jh in slabs():
                Y += Jh.T @ (Jh @ W)
            return np.asfortranarray(Y / ns_total)
        return apply_A
    if args.workload == "pod":
        n = wl.n

        def slabs():
            if world == 1:
                for _, cnt, Xh in _stream_rows(wl.X, 512):
                    yield Xh
            else:
                for first in range(0, n, 512):
                    part = workloads.pod_workload(wl.N, n, latent=256, rate=0.05, seed=3, first_snapshot=first, n_local=min(512, n - first))
                    yield part.X.to_vectors()

        def apply_A(W):
            Y = np.zeros((wl.N, W.shape[1]))
            for Xh in slabs():
                Y += Xh.T @ (Xh @ W)
            return np.asfortranarray(Y / n)
        return apply_A
    M = wl.M

    def apply_A(W):                                          # M C M, the covariance streamed by rows (C is symmetric)
        MW = M @ W
        CMW = np.empty_like(MW)
        for first, cnt, Ch in _stream_rows(wl.C, 4000):
            CMW[first:first + cnt] = Ch @ MW
        return np.asfortranarray(M @ CMW)
    return apply_A


def host_reference(args, hf, wl, prior, world, Omega_host, r, hp_o):
    apply_A = dense_streaming_operator(args, hf, wl, world)
    if args.workload == "kle":
        import scipy.sparse.linalg as spla
        lu = spla.splu(wl.M.tocsc())
        return hp_o.double_pass_blas3(apply_A, Omega_host, r, apply_B=lambda W: wl.M @ W,
                                      apply_Binv=lambda W: np.asfortranarray(lu.solve(np.ascontiguousarray(W)))), (lambda W: wl.M @ W)
    if prior is not None:
        return hp_o.double_pass_blas3(apply_A, Omega_host, r, apply_B=lambda W: prior.R @ W,
                                      apply_Binv=lambda W: np.asfortranarray(prior.Rsolver.solve_block(W))), (lambda W: prior.R @ W)
    return hp_o.double_pass_blas3(apply_A, Omega_host, r), None


def _timed(fn, *a):
    t0 = time.perf_counter()
    out = fn(*a)
    return out, time.perf_counter() - t0


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_topology():
    """Hardware threads, physical cores and sockets of this host from /proc/cpuinfo ((physical id, core id) pairs)."""
    cores, sockets = set(), set()
    phys = core = None
    try:
        for line in list(open("/proc/cpuinfo")) + [""]:
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                    sockets.add(phys)
                phys = core = None
    except OSError:
        pass
    threads = os.cpu_count() or 1
    return {"threads": threads, "physical_cores": len(cores) or threads, "sockets": len(sockets) or 1}


def _cpu_quota(cpu_max_path="/sys/fs/cgroup/cpu.max", v1_dir="/sys/fs/cgroup/cpu"):
    """CPUs' worth of run time the container's cgroup grants this process (cpu.max: quota / period), or None without a limit.
    The GPU boxes of this pool report 256 hardware threads and an affinity mask of 256 with cpu.max = 1600000 100000: SIXTEEN CPUs.
    A process that makes more threads runnable than that (numpy's default BLAS pool after any host matmul, busy-waiting helpers)
    is throttled by the scheduler for the rest of the 100 ms period -- every thread of it, the one that feeds the GPU included:
    the 15-80 ms stalls of rounds 4-5 (profiles/r06_eig_stall_diagnosis.txt)."""
    for path in (cpu_max_path,):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                return max(1, int(round(float(quota) / float(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open(os.path.join(v1_dir, "cpu.cfs_quota_us")).read())
        per = int(open(os.path.join(v1_dir, "cpu.cfs_period_us")).read())
        if q > 0 and per > 0:
            return max(1, int(round(q / per)))
    except (OSError, ValueError):
        pass
    return None


_BLAS_LIMIT = None


def limit_host_blas_to_cpu_quota():
    """Keep numpy's BLAS pool within what the cgroup lets the process run (see _cpu_quota): process-wide, for the life of it."""
    global _BLAS_LIMIT
    quota = _cpu_quota()
    if quota is None or _BLAS_LIMIT is not None:
        return quota
    try:
        from threadpoolctl import threadpool_limits
        _BLAS_LIMIT = threadpool_limits(limits=max(1, min(os.cpu_count() or 1, quota)))
    except Exception:
        pass
    return quota


def cpu_baseline(args, wl, prior, Omega_host, r, N, hp_o, hf_o):
    """The reference's CPU path on the GPU box's host cores, on bounded samples scaled to the full workload (every
    scaling factor is stated).  Two legs, each at all threads and at one thread:
      reference_style -- what hippylib executes for hippyflow's operators: the operator applied one Omega column at a
                         time (MatMvMult's fallback loop, collectiveOperator.py:31-38, over the in-tree numpy statement
                         MeanJTJfromDataOperator.mult / LowRankOperator), column-by-column MGS with re-orthogonalisation,
                         numpy eigh, MvDSmatMult;
      blas3           -- the best-effort CPU form: block applies as threaded GEMMs, Householder QR."""
    from threadpoolctl import threadpool_limits
    topo = _cpu_topology()
    quota = _cpu_quota()
    # what the process can actually run at once: the hardware threads, or the cgroup's CPU quota where that is smaller (a BLAS pool
    # beyond the quota is throttled by the scheduler, not faster)
    cores = topo["threads"] if quota is None else max(1, min(topo["threads"], quota))
    # BLAS thread counts: every hardware thread (what a default numpy does), the physical cores of ONE socket (level-1/2
    # style loops like the reference's column-by-column MGS lose to their own fork/join and cross-socket traffic beyond that:
    # 0.81 s at 256 threads against 0.027 s at one, BENCH_r04), and one thread.  The best of each leg is reported.
    per_socket = max(1, min(cores, topo["physical_cores"] // max(1, topo["sockets"])))
    settings = [("threads_all", cores), ("threads_one_socket_cores", per_socket), ("threads_1", 1)]
    if per_socket == cores:          # (a CPU quota below one socket's cores: the two settings coincide)
        settings = [st for st in settings if st[0] != "threads_one_socket_cores"]
    quick = args.cpu_baseline == "quick"
    # quick (the extra workloads of a default run): the reference-style leg at one thread only (its level-1 loops are fastest
    # there on every box seen), the BLAS-3 leg at all threads and at one socket's cores
    do_ref = {"threads_all": not quick, "threads_one_socket_cores": not quick, "threads_1": True}
    do_blas3 = {"threads_all": True, "threads_one_socket_cores": True, "threads_1": not quick}
    k = Omega_host.shape[1]
    legs = {"reference_style": {}, "blas3": {}}
    notes = {}
    for label, limit in settings:
        with threadpool_limits(limits=limit):
          if do_ref[label]:
              # ------------------------------------------------------------ reference-style leg, on a reduced problem
              N_s = max(1000, N // 4)
              k_s = min(8, k)
              W_s = np.ascontiguousarray(Omega_host[:N_s, :k_s])
              Z = hp_o.as_block(Omega_host[:N_s])
              if args.workload == "as":
                  ns_s = max(1, min(8, wl.J.nvec() // wl.q))
                  J_s = wl.J.view(0, ns_s * wl.q).to_vectors()[:, :N_s].reshape(ns_s, wl.q, N_s).copy()
                  op = hf_o.MeanJTJOperator(J_s)
                  units, units_s = wl.ns_total, ns_s
              elif args.workload == "pod":
                  n_s = 64
                  X_s = wl.X.view(0, n_s).to_vectors()[:, :N_s].copy()
                  op = hf_o.SnapshotGramOperator(X_s)
                  units, units_s = wl.n, n_s
              else:
                  rows_s = min(500, N)
                  C_s = wl.C.view(0, rows_s).to_vectors()[:, :N_s].copy()

                  class _Rows:                       # rows_s rows of y = C x per call (dense mat-vec, as npToDolfinOperator.mult)
                      def mult(self, x, y):
                          y[:rows_s] = C_s @ x
                  op = _Rows()
                  units, units_s = N, rows_s
              y = np.zeros(N_s)

              def ref_apply():
                  for j in range(k_s):
                      op.mult(W_s[:, j], y)
              _, t_apply = _timed(ref_apply)
              _, t_mgs = _timed(hp_o.mgs_reortho, Z)
              T = Z.T @ Z
              _, t_eig = _timed(np.linalg.eigh, T)
              U_s = hp_o.new_block(N_s, r)
              _, t_back = _timed(hp_o.mv_ds_mat_mult, Z, np.ascontiguousarray(T[:, :r]), U_s)
              fN = N / N_s
              full_apply = t_apply * (k / k_s) * (units / units_s) * fN       # columns x operator rows x vector length
              t_ref = 2.0 * full_apply + (t_mgs + t_back) * fN + t_eig
              legs["reference_style"][label] = {"seconds_full_estimate": t_ref, "value": N * r / t_ref / 1e9,
                                              "scale_factors": {"apply": (k / k_s) * (units / units_s) * fN, "N": fN},
                                                "measured_seconds": {"apply": t_apply, "mgs_reortho": t_mgs, "eigh": t_eig, "MvDSmatMult": t_back}}
              notes["reference_style"] = ("operator applied to %d of %d Omega columns on %d of %d %s at N/%d = %d rows, scaled linearly in "
                                          "columns, %s and N, x2 applications; MGS with re-orthogonalisation and MvDSmatMult at N/%d scaled "
                                          "linearly in N; eigh at full k=%d%s"
                                          % (k_s, k, units_s, units, {"as": "samples", "pod": "snapshots", "kle": "covariance rows"}[args.workload],
                                             N // N_s, N_s, {"as": "samples", "pod": "snapshots", "kle": "rows"}[args.workload], N // N_s, k,
                                             "; B / B^-1 applications not included" if (prior is not None or args.workload == "kle") else ""))
          if do_blas3[label]:
              # ------------------------------------------------------------ BLAS-3 leg, full N, bounded operator sample
              if args.workload == "as":
                  ns_b = max(1, min(16, wl.J.nvec() // wl.q))     # this rank's share may hold fewer than 16 samples (many ranks, few samples)
                  Jh = wl.J.view(0, ns_b * wl.q).to_vectors()
                  _, t_apply = _timed(lambda: Jh.T @ (Jh @ Omega_host))
                  t_apply *= wl.ns_total / ns_b
                  note_b = "mean-JtJ apply as two GEMMs on %d of %d samples at full N=%d, k=%d, scaled linearly in samples, x2" % (ns_b, wl.ns_total, N, k)
              elif args.workload == "pod":
                  n_b = 256
                  Xh = wl.X.view(0, n_b).to_vectors()
                  _, t_apply = _timed(lambda: Xh.T @ (Xh @ Omega_host))
                  t_apply *= wl.n / n_b
                  note_b = "snapshot-Gram apply as two GEMMs on %d of %d snapshots at full N, scaled linearly, x2" % (n_b, wl.n)
              else:
                  rows_b = min(4000, N)
                  Ch = wl.C.view(0, rows_b).to_vectors()
                  _, t_c = _timed(lambda: Ch @ Omega_host)
                  _, t_m = _timed(lambda: wl.M @ Omega_host)
                  t_apply = t_c * (N / rows_b) + 2.0 * t_m
                  note_b = "M C M apply: dense C rows GEMM on %d of %d rows scaled linearly + 2 sparse M products, x2" % (rows_b, N)
              extra = 0.0
              if args.workload == "kle":
                  import scipy.sparse.linalg as spla
                  lu, t_fac = _timed(lambda: spla.splu(wl.M.tocsc()))
                  _, t_sol = _timed(lambda: lu.solve(np.ascontiguousarray(Omega_host)))
                  _, t_bq = _timed(lambda: hp_o._borth_blas3(Omega_host.copy(order="F"), lambda W: wl.M @ W))
                  extra = t_fac + t_sol + t_bq
                  note_b += "; + splu(M) factorisation, one block solve and the M-orthogonal QR (Householder + 2 Cholesky-QR rounds) at full size"
              elif prior is not None:
                  _, t_sol = _timed(lambda: prior.Rsolver.solve_block(Omega_host))
                  _, t_bq = _timed(lambda: hp_o._borth_blas3(Omega_host.copy(order="F"), lambda W: prior.R @ W))
                  extra = t_sol + t_bq
                  note_b += "; + one R^-1 block solve (factorisation of A not counted: the prior owns it) and the R-orthogonal QR at full size"
              else:
                  _, extra = _timed(lambda: hp_o._qr_posdiag(Omega_host))
                  note_b += "; + Householder QR at full size"
              T = Omega_host[:k].T @ Omega_host[:k]
              _, t_eig = _timed(np.linalg.eigh, T)
              _, t_back = _timed(lambda: Omega_host @ T[:, :r])
              t_b = 2.0 * t_apply + extra + t_eig + t_back
              legs["blas3"][label] = {"seconds_full_estimate": t_b, "value": N * r / t_b / 1e9}
              notes["blas3"] = note_b + "; eigh and U = Q V at full size"
    thread_count = dict(settings)
    best_label = max(legs["blas3"], key=lambda lb: legs["blas3"][lb]["value"])
    best = legs["blas3"][best_label]
    best_ref_label = max(legs["reference_style"], key=lambda lb: legs["reference_style"][lb]["value"])
    # each component of the reference-style leg at the thread count that suits IT (a user would pin the level-1 loops to one
    # thread and leave the mat-vecs threaded; the reference itself runs one MPI rank per core): the best composed estimate
    comp = {}
    for lb, leg in legs["reference_style"].items():
        for name, sec in leg["measured_seconds"].items():
            comp[name] = min(comp.get(name, float("inf")), sec)
    ref_scale = legs["reference_style"][best_ref_label]["scale_factors"]
    t_comp = 2.0 * comp["apply"] * ref_scale["apply"] + (comp["mgs_reortho"] + comp["MvDSmatMult"]) * ref_scale["N"] + comp["eigh"]
    return {"value": best["value"], "unit": "GDoF*rank/s", "cores": thread_count[best_label], "kind": "port", "cpu_model": _cpu_model(),
            "threads": topo["threads"], "physical_cores": topo["physical_cores"], "sockets": topo["sockets"],
            "cgroup_cpu_quota": quota, "thread_settings": thread_count, "best_setting": best_label,
            "sample": "blas3 leg at %s = %d BLAS threads (the best of %s): %s" % (best_label, thread_count[best_label], ", ".join(thread_count), notes["blas3"]),
            "seconds_full_estimate": best["seconds_full_estimate"],
            # the reference-style leg: one entry per thread setting (each a run that a user can reproduce), the best of those, and --
            # named for what it is -- a COMPOSITE no single run achieves: every component at the thread setting that suits it
            "reference_style": dict(legs["reference_style"], sample=notes["reference_style"], best_single_setting=best_ref_label,
                                    best_single_setting_value=legs["reference_style"][best_ref_label]["value"],
                                    best_single_setting_seconds_full_estimate=legs["reference_style"][best_ref_label]["seconds_full_estimate"],
                                    composite_of_per_component_best_settings={
                                        "value": max(legs["reference_style"][best_ref_label]["value"], N * r / t_comp / 1e9),
                                        "seconds_full_estimate": min(legs["reference_style"][best_ref_label]["seconds_full_estimate"], t_comp),
                                        "per_component_seconds": comp,
                                        "note": "MGS / small products at their best thread count, mat-vecs at theirs: not one run"}),
            "blas3": dict(legs["blas3"], sample=notes["blas3"])}


# ---------------------------------------------------------------------------------------------- the benchmark
def dipnet_line(args):
    """Config 5: r_in = 50 AS x r_out = 50 POD projected residual network, bf16 training on one MI355X, relative l2 test
    error next to the fp32 CPU run of the same PyTorch restatement (keras parity is unpinned: TensorFlow cannot run here)."""
    import tempfile
    from hippyflow_amd import workloads
    from hippyflow_amd import surrogate
    scale = 4 if args.quick else 1
    wl = workloads.dipnet_workload(dM=20000 // scale, dQ=400, hidden=80, n_train=8192 // scale, n_test=1024, ns=64)
    with tempfile.TemporaryDirectory() as tmp:
        epochs = max(1, args.steps * 60)                 # default --steps 10: the tuned 600-epoch cosine schedule (27 s on the GPU)
        res = surrogate.run_config5(wl, tmp, r_in=50, r_out=50, epochs=epochs, batch_size=256)
    out = {"metric": "projected-network surrogate training throughput (samples/s) + relative l2 test error", "value": res["gpu_samples_per_second"],
           "unit": "samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "bf16 autocast (fp32 parameters and optimiser state)", "data": "synthetic nonlinear map q = W2 tanh(W1^T m)",
           "config": {"workload": "config5 DIPNet: ProjectedLowRankResidualNetwork, r_in=50 (device AS solve) x r_out=50 (device POD solve), "
                                  "dM=%d, dQ=%d, %d training points, %d epochs of Adam (cosine schedule)" % (wl.dM, wl.dQ, wl.m_train.shape[0], epochs)},
           "result": res}
    print(json.dumps(out), flush=True)


def gpu_sysfs_snapshot(pci_bus_id=None):
    """Shader clock / power the driver exposes through sysfs (plain file reads; no child process, no GPU call): the big
    contractions are power-limited, so the clock they were given is part of the record.  Best effort: {} where the files
    are not there."""
    import glob
    snap = {}
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        try:
            if pci_bus_id and pci_bus_id.lower() not in os.path.realpath(dev).lower():
                continue
            rec = {}
            try:
                for line in open(os.path.join(dev, "pp_dpm_sclk")).read().splitlines():
                    if line.strip().endswith("*"):
                        rec["sclk"] = line.split(":", 1)[1].replace("*", "").strip()
            except OSError:
                pass
            for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
                for name, key, scale in (("power1_average", "power_w", 1e-6), ("power1_input", "power_w", 1e-6),
                                         ("freq1_input", "sclk_hz", 1.0), ("temp1_input", "temp_c", 1e-3)):
                    try:
                        rec.setdefault(key, float(open(os.path.join(hw, name)).read().strip()) * scale)
                    except (OSError, ValueError):
                        pass
            if rec:
                snap[os.path.basename(os.path.dirname(dev))] = rec
        except OSError:
            continue
    return snap


def ingest_line(args):
    """--ingest: host -> HBM rate of the sample-by-sample filling of a Jacobian block (the reference's sampling loop,
    activeSubspaceProjector.py:178-221 / PODProjector.py:343-357).  Never part of `value`: the solve is timed
    with its operator data resident.  Three legs on config-4-shaped samples (q x N doubles each): the synchronous upload
    from pageable memory (hfmi_block_upload), the asynchronous one from pinned double buffers (hfmi_block_upload_async)
    with the GPU idle, and the same while a mean-J^T J application of resident samples runs on the compute stream."""
    import hippyflow_amd as hf
    from hippyflow_amd import _lib as L
    ctx = hf.Context.default()
    q, N = 100, 200000 // (8 if args.quick else 1)
    ns = 24
    sample_bytes = q * N * 8
    rng = np.random.default_rng(0)
    src = rng.standard_normal((q, N))
    blk = hf.MultiVector(N, ns * q)
    # (a) pageable, synchronous
    t0 = time.perf_counter()
    for i in range(ns):
        v = blk.view(i * q, q)                    # (kept alive across the call: the handle dies with the view object)
        L.call("hfmi_block_upload", v.handle, L.ptr(src), L.LAYOUT_VECTORS)
    ctx.synchronize()
    t_pageable = time.perf_counter() - t0
    # (b) pinned, asynchronous, producer = a copy into the pinned buffer (a stand-in for the host PDE solve's output)
    def stream():
        for _ in range(ns):
            yield src
    t0 = time.perf_counter()
    blk2 = hf.ingest_stream(stream(), ns, q, N)
    ctx.synchronize()
    t_pinned = time.perf_counter() - t0
    np.testing.assert_array_equal(blk2.view(0, q).to_vectors(), src)
    # (b') the DMA alone: the producer writes in place into the pinned buffers (no host copy in the loop)
    pins = [hf.pinned_empty((q, N)) for _ in range(2)]
    for pb in pins:
        pb[...] = src
    t0 = time.perf_counter()
    tk = [None, None]
    for i in range(ns):
        if tk[i & 1] is not None:
            ctx.ingest_wait(tk[i & 1])
        tk[i & 1] = blk2.view(i * q, q).upload_async(pins[i & 1])
    ctx.ingest_fence()
    ctx.synchronize()
    t_dma = time.perf_counter() - t0
    # (c) the same with the compute stream busy
    resident = hf.MultiVector(N, 32 * q)
    hf.parRandom.normal(1.0, resident)
    op = hf.MeanJTJfromDataOperator.from_block(resident, 32, q)
    W, Y = hf.MultiVector(N, 74), hf.MultiVector(N, 74)
    hf.parRandom.normal(1.0, W)
    hf.MatMvMult(op, W, Y)
    ctx.synchronize()
    t0 = time.perf_counter()
    hf.MatMvMult(op, W, Y)
    ctx.synchronize()
    t_apply_alone = time.perf_counter() - t0
    reps = max(1, int(t_pinned / max(t_apply_alone, 1e-4)) + 1)
    t0 = time.perf_counter()
    for _ in range(reps):
        hf.MatMvMult(op, W, Y)                      # enqueued; the host goes on to produce samples
    blk3 = hf.ingest_stream(stream(), ns, q, N)
    t_ingest_busy = time.perf_counter() - t0
    ctx.synchronize()
    t_both = time.perf_counter() - t0
    out = {"metric": "host -> HBM ingest rate of sample-by-sample block filling (GB/s)", "unit": "GB/s",
           "value": ns * sample_bytes / t_pinned / 1e9, "n_gpus": 1, "higher_is_better": True, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "config-4-shaped Jacobian samples, %d x (%d x %d) doubles = %.2f GB" % (ns, q, N, ns * sample_bytes / 1e9)},
           "pageable_synchronous_gbs": ns * sample_bytes / t_pageable / 1e9,
           "pinned_async_gbs": ns * sample_bytes / t_pinned / 1e9,
           "pinned_async_in_place_producer_gbs": ns * sample_bytes / t_dma / 1e9,
           "pinned_async_gbs_while_computing": ns * sample_bytes / t_ingest_busy / 1e9,
           "compute_alone_ms": t_apply_alone * 1e3, "compute_reps_alongside": reps,
           "compute_plus_ingest_ms": t_both * 1e3, "serial_sum_ms": (reps * t_apply_alone + t_pinned) * 1e3,
           "note": "never part of the solve's `value`; the producer here is a host memcpy into the pinned buffer"}
    del blk3
    print(json.dumps(out), flush=True)


def kernel_point_line(args):
    """--kernel-point: the north star's kernel point G = X^T Omega (snapshot^T x probe block) at N = 1e6, k = r + p = 138 for
    several snapshot counts n: achieved algorithmic GB/s against the 8 TB/s HBM roof AND TFLOP/s against the 78.6 TFLOP/s fp64
    MFMA roof, the binding one named.  Median of five 10-launch batches per variant, best of the two operand
    orientations / kernels, timed with HIP events on the library's stream (hfmi_bench_tsgemm_tn)."""
    import ctypes as C
    import hippyflow_amd as hf
    from hippyflow_amd import _lib as L
    hf.Context.default()
    N, k = (1000000 // (8 if args.quick else 1)), 138
    W = hf.MultiVector(N, k)
    hf.parRandom.normal(1.0, W)
    rows = []
    for n in (8, 16, 32, 48, 64, 96, 138, 512, 2048):
        X = hf.MultiVector(N, n)
        hf.parRandom.normal(1.0, X)
        best = None
        for ss in ((1, 0) if n <= 160 else (0,)):     # 1: the skinny kernel where applicable (the default), 0: force tsgemm_tn
            L.call("hfmi_tuning_set", b"ss", ss)
            for orient in ("X^T W", "(W^T X)^T"):
                A, B = (X, W) if orient == "X^T W" else (W, X)
                reps = []
                for _ in range(5):
                    ms = C.c_double(0)
                    L.call("hfmi_bench_tsgemm_tn", A.handle, B.handle, 0, 10, None, C.byref(ms))
                    reps.append(ms.value)
                t_med = float(np.median(reps))
                if best is None or t_med < best[1]:
                    best = (orient + (" [ss]" if ss and n <= 160 else " [tn]"), t_med)
        L.call("hfmi_tuning_set", b"ss", 1)
        ts = best[1] * 1e-3
        by, fl = 8.0 * (N * n + N * k + n * k), 2.0 * N * n * k
        ai = fl / by
        bound = "mfma" if ai > FP64_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9) else "hbm"
        rows.append({"n": n, "kernel": best[0], "ms": best[1], "hbm_frac": by / ts / (HBM_PEAK_GBS * 1e9),
                     "mfma_frac": fl / ts / (FP64_MFMA_PEAK_TFLOPS * 1e12), "bound": bound,
                     "frac": (fl / ts / (FP64_MFMA_PEAK_TFLOPS * 1e12)) if bound == "mfma" else by / ts / (HBM_PEAK_GBS * 1e9)})
        del X
    # the same shapes against what THIS box delivers in THIS job: a read-only 16-byte stream and the MFMA loop on Gaussian operands.
    # At the ridge (n = 48 ... 138) both are loaded at once: frac_of_measured = max(bytes / read rate, flops / MFMA rate) / time
    measured = {}
    try:
        ctx = hf.Context.default()
        measured.update(ctx.bench_hbm_read())
        measured.update(ctx.bench_random_peaks())
        for row in rows:
            by, fl = 8.0 * (N * row["n"] + N * k + row["n"] * k), 2.0 * N * row["n"] * k
            t_roof = max(by / (measured["hbm_read_gbs"] * 1e9), fl / (measured["mfma_f64_tflops_random_operands"] * 1e12))
            row["frac_of_measured"] = t_roof / (row["ms"] * 1e-3)
            row["bound_measured"] = "hbm" if by / (measured["hbm_read_gbs"] * 1e9) >= fl / (measured["mfma_f64_tflops_random_operands"] * 1e12) else "mfma"
    except Exception as exc:
        measured["error"] = str(exc)
    print(json.dumps({"kernel_point": {"N": N, "k": k, "rows": rows, "build_tag": hf.build_tag(),
                                       "peaks": {"hbm_gbs": HBM_PEAK_GBS, "fp64_mfma_tflops": FP64_MFMA_PEAK_TFLOPS},
                                       "peaks_measured_in_job": measured}}), flush=True)


def eig_large_line(args):
    """--eig-large: la.eigh(G) of the deterministic POD for 256 < n <= 8192 snapshots on the device: Gram matrix of
    n + 50 decaying snapshots, all eigenvectors, host matrix in and out (min of three calls after one warm-up), the POD form beside it
    (hfmi_block_gram_eig: X^T X formed on the device, 128 leading eigenvectors returned), numpy.linalg.eigh on the host's threads as the
    CPU baseline (not from n = 4096, where it takes 6-7 s; 37 s at 8192) and the parity of the two: eigenvalues, orthonormality,
    residual (at n = 8192: two calls instead of three, the checks on the 256 leading eigenvectors)."""
    import hippyflow_amd as hf
    hf.Context.default()
    rng = np.random.default_rng(0)
    rows = []
    for n in ((300, 512) if args.quick else (512, 1024, 2048, 4096, 8192)):
        X = rng.standard_normal((n, n + 50)) * np.exp(-0.01 * np.arange(n + 50))[None, :]
        G = X @ X.T
        hf.sym_eig_small(G)
        ts, reps = [], (3 if n <= 4096 else 2)
        for _ in range(reps):
            t0 = time.perf_counter()
            d, V = hf.sym_eig_small(G)
            ts.append(time.perf_counter() - t0)
        Xm = hf.MultiVector.from_vectors(X)             # n snapshots of length n + 50: one per vector
        Xm.gram_eig(Xm, min(128, n))
        tg = []
        for _ in range(reps):
            t0 = time.perf_counter()
            dg, Vg = Xm.gram_eig(Xm, min(128, n))
            tg.append(time.perf_counter() - t0)
        nchk = n if n <= 4096 else 256
        row = {"n": n, "ms": 1e3 * min(ts), "ms_max": 1e3 * max(ts), "gram_eig_128_ms": 1e3 * min(tg),
               "orthonormality": float(np.abs(V[:, :nchk].T @ V[:, :nchk] - np.eye(nchk)).max()),
               "residual_rel": float(np.abs(G @ V[:, :nchk] - V[:, :nchk] * d[:nchk]).max() / d[0]), "checked_eigenvectors": nchk,
               "host_eigh_ms": None, "eig_err_rel_vs_host": None,
               "gram_eig_err_rel": float(np.abs(dg - d).max() / d[0])}
        if n <= 2048:
            t0 = time.perf_counter()
            w = np.linalg.eigh(G)[0]                  # eigenvalues AND eigenvectors, as the reference's la.eigh(G) computes them
            row["host_eigh_ms"] = 1e3 * (time.perf_counter() - t0)
            row["eig_err_rel_vs_host"] = float(np.abs(d - w[::-1]).max() / w[-1])
        rows.append(row)
    topo = _cpu_topology()
    print(json.dumps({"eig_large": {"rows": rows, "dtype": "f64", "build_tag": hf.build_tag(), "host_threads": topo["threads"],
                                    "what": "hfmi_sym_eig_small (all eigenvectors, host in / host out) and hfmi_block_gram_eig (POD form) "
                                            "vs numpy.linalg.eigh"}}), flush=True)


def _compact(line):
    """What an extra workload contributes to the headline line: time, dominant kernel against its roof, parity, a host baseline."""
    out = {k: line.get(k) for k in ("value", "unit", "ms_per_step", "median_ms_per_step", "step_ms_min_max", "value_from_median",
                                    "literal_T_ms_per_step", "steps", "warmup")}
    out["workload"] = (line.get("config") or {}).get("workload")
    rf = line.get("roofline") or {}
    out["roofline"] = {k: rf.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "traffic")}
    out["parity"] = {k: v for k, v in (line.get("parity") or {}).items() if k not in ("note", "oracle_form")}
    out["phases_ms_per_step"] = line.get("phases_ms_per_step")
    cb = line.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "kind", "cores", "threads", "physical_cores", "sockets", "cgroup_cpu_quota",
                                                       "best_setting", "seconds_full_estimate", "sample")}
        rs = cb.get("reference_style") or {}
        out["cpu_baseline"]["reference_style_best_single_setting_value"] = rs.get("best_single_setting_value")
        out["cpu_baseline"]["reference_style_composite_of_per_component_best_settings_value"] = (
            rs.get("composite_of_per_component_best_settings") or {}).get("value")
    comm = line.get("communicator") or {}
    if comm.get("ranks"):
        out["communicator"] = comm
    if line.get("sptrsv_feasibility"):
        out["sptrsv_feasibility"] = line["sptrsv_feasibility"]
        out["host_solver"] = (line.get("config") or {}).get("Binv")
    return out


def run_extras(args):
    """The rest of the evidence set, appended to the driver's one default run: each extra is a CHILD
    process of this one (a fresh interpreter started with subprocess: nothing is re-exec'ed; the headline workload has been
    freed), bounded by its own time-out, so that a failure or a hang there can never cost the headline line.  Budget: about
    90 s of wall clock in total."""
    import subprocess
    q = ["--quick"] if args.quick else []
    jobs = [("kernel_point", ["--kernel-point"] + q, 60),
            ("eig_large", ["--eig-large"] + q, 90),
            ("config3", ["--workload", "pod", "--steps", "10", "--warmup", "3", "--cpu-baseline", "quick"] + q, 120),
            ("config2", ["--workload", "kle", "--steps", "4", "--warmup", "1", "--cpu-baseline", "quick"] + q, 180),
            ("shard64", ["--samples-total", "64", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"] + q, 90),
            ("shard64_rccl_1rank", ["--samples-total", "64", "--dist-single", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"] + q, 90),
            # the reference's DEFAULT active-subspace path (construct_input_subspace(prior_preconditioned=True), activeSubspaceProjector.py:400,
            # :447-450: doublePassG(A, prior.R, prior.Rsolver)): the 64-sample shard with B = R as CSR on the device and B^-1 = the host
            # sparse-LU pool (one worker process per probe vector, at most the box's physical cores and the cgroup's CPU quota), pinned-slab overlap on
            ("as_prior_shard64", ["--prior", "--samples-total", "64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--host-workers",
                                  str(max(1, min(74, _cpu_topology()["physical_cores"], _cpu_quota() or 74)))] + q, 150)]
    extras, seconds = {}, {}
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    for name, argv, limit in jobs:
        t0 = time.perf_counter()
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--headline-only"] + argv, env=env, stdout=subprocess.PIPE,
                                 stderr=subprocess.PIPE, timeout=limit)
            lines = [ln for ln in res.stdout.decode("utf-8", "replace").splitlines() if ln.startswith("{")]
            if res.returncode != 0 or not lines:
                extras[name] = {"error": "exit code %d" % res.returncode, "stderr_tail": res.stderr.decode("utf-8", "replace")[-600:]}
            else:
                line = json.loads(lines[-1])
                extras[name] = line[name] if name in ("kernel_point", "eig_large") else _compact(line)
        except subprocess.TimeoutExpired:
            extras[name] = {"error": "timed out after %d s" % limit}
        except Exception as exc:                          # never let an extra cost the headline
            extras[name] = {"error": repr(exc)}
        seconds[name] = time.perf_counter() - t0
    extras["extras_wall_seconds"] = seconds
    return extras


def _error_line(args, message, **more):
    out = {"metric": "randomized-SVD throughput (GDoF*rank/s)", "value": None, "unit": "GDoF*rank/s", "n_gpus": args.gpus,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": args.workload}, "error": message}
    out.update(more)
    return out


def spawn_and_report(args):
    """Plain ``python bench.py --gpus N``: this parent touches no GPU; it starts N fresh child interpreters (one per device,
    nothing is re-exec'ed), captures their output and prints exactly ONE JSON line: rank 0's on success, otherwise an error
    line with every rank's exit code, the transport the ranks had agreed on (if they got that far) and the tail of every
    rank's stderr.  A rank that dies takes its peers down at once; a hang ends at the time-out (HFMI_BENCH_TIMEOUT_S)."""
    from hippyflow_amd.launch import spawn_ranks
    report = {}
    limit = float(os.environ.get("HFMI_BENCH_TIMEOUT_S", "1800"))
    rc = spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus, timeout=limit, report=report)
    lines = [ln for ln in report.get("stdout_rank0", "").splitlines() if ln.startswith("{")]
    for rnk, tail in sorted(report.get("stderr_tail", {}).items()):
        if tail and (rc != 0 or rnk == "0"):
            sys.stderr.write("---- rank %s stderr (tail) ----\n%s\n" % (rnk, tail))
    if rc == 0 and lines:
        print(lines[-1], flush=True)
        return 0
    transport = None
    for tail in report.get("stderr_tail", {}).values():
        for ln in tail.splitlines():
            if ln.startswith("[bench] communicator "):
                try:
                    transport = json.loads(ln[len("[bench] communicator "):])
                except ValueError:
                    pass
    why = ("timed out after %.0f s" % limit) if report.get("timed_out") else \
          ("rank %s exited with code %s" % (report.get("first_failed"), (report.get("codes") or [None])[report.get("first_failed") or 0]))
    if rc == 0:
        why, rc = "rank 0 printed no JSON line", 1
    print(json.dumps(_error_line(args, why, rank_exit_codes=report.get("codes"), ranks_stopped_by_launcher=report.get("stopped"),
                                 communicator=transport or "not reached", launcher_seconds=report.get("seconds"),
                                 stderr_tail=report.get("stderr_tail"))), flush=True)
    return rc or 1


def main():
    args = parse_args()
    limit_host_blas_to_cpu_quota()
    if args.workload == "dipnet":
        return dipnet_line(args)
    if args.kernel_point:
        return kernel_point_line(args)
    if args.eig_large:
        return eig_large_line(args)
    if args.ingest:
        return ingest_line(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_and_report(args))
    # stdout carries exactly ONE JSON line: anything libraries print (e.g. the RCCL version banner) goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    try:
        out = solve_line(args, world, rank)
    except BaseException as exc:
        # under an external launcher (python -m torch.distributed.run) there is no parent of ours to report: rank 0 leaves ONE
        # JSON line that says what happened, every rank exits non-zero
        if rank == 0 and not isinstance(exc, KeyboardInterrupt):
            import traceback
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            print(json.dumps(_error_line(args, "%s: %s" % (type(exc).__name__, exc), traceback=traceback.format_exc()[-1500:])), flush=True)
            os.dup2(2, 1)
        if isinstance(exc, SystemExit) and exc.code in (0, None):
            raise SystemExit(1)
        raise
    if out is None:                                  # ranks other than 0
        return
    if kernel_extras_wanted(args, world):
        out.update(run_extras(args))
        out["extra_keys"] = ["kernel_point", "eig_large", "config3", "config2", "shard64", "shard64_rccl_1rank", "as_prior_shard64"]
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)                      # whatever the libraries print while shutting down stays off stdout


def kernel_extras_wanted(args, world):
    """The extra evidence lines ride on the driver's default run only: config 4, one GPU, all 512 samples, plain doublePass."""
    return (world == 1 and not args.headline_only and args.workload == "as" and not args.prior and not args.dist_single
            and args.samples_total == 512)


def solve_line(args, world, rank):
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if args.transport != "auto":
        os.environ["HFMI_COMM_TRANSPORT"] = args.transport
    if args.prior and args.workload != "as":
        raise SystemExit("--prior belongs to the config-4 workload (--workload as)")

    import hippyflow_amd as hf
    if hf.device_count() < 1:
        raise SystemExit("bench.py needs a GPU (libhfmi has no CPU path)")
    ctx = hf.Context.default()                       # cuda:LOCAL_RANK
    use_dist = world > 1 or args.dist_single
    if world > 1:
        collective = hf.NativeCollective.from_env(ctx)          # id through $HFMI_COMM_ID_FILE / the launcher's temp file
        try:                                                    # what the ranks agreed on: the launcher's error line quotes it
            sys.stderr.write("[bench] communicator %s\n" % json.dumps(dict(collective.describe(), rank=rank)))
            sys.stderr.flush()
        except Exception:
            pass
    elif args.dist_single:
        collective = hf.NativeCollective.from_unique_id(hf.NativeCollective.unique_id(), 1, 0, ctx=ctx)
    else:
        collective = hf.NullCollective()

    wl, op, B, Binv, prior, N, r, p, desc = build_workload(args, hf, rank, world)
    k = r + p
    A = hf.CollectiveOperator(op, collective, mpi_op="avg") if use_dist else op
    hf.parRandom.reseed(1)
    Omega = hf.MultiVector(N, k)
    hf.parRandom.normal(1.0, Omega)          # identical on every rank (counter-based RNG): no broadcast

    def step(**kw):
        if B is None:
            return hf.doublePass(A, Omega, r, s=1, **kw)
        return hf.doublePassG(A, B, Binv, Omega, r, s=1, **kw)

    def barrier():
        collective.barrier()                         # drains this rank's stream, then meets the other ranks
        ctx.synchronize()

    for _ in range(args.warmup):
        d, U = step()
    barrier()
    pci = ctx.pci_bus_id()
    sysfs_before = gpu_sysfs_snapshot(pci) if rank == 0 else {}
    # Inside the timed region only the contractions of at least 2 Gflop are bracketed by events (what `roofline` needs): every
    # event pair between two dependent kernels costs ~5 us of idle GPU, and with one pair per contraction and per phase the
    # 64-sample shard step carried ~32 of them (30-70 us per step, scripts/prof_level_ab.py).  The per-phase breakdown comes from
    # extra, untimed steps below.  HFMI_BENCH_PROF_LEVEL=2 puts everything back inside the timed region (A/B).
    from hippyflow_amd import _lib as _L
    timed_level = int(os.environ.get("HFMI_BENCH_PROF_LEVEL", "1"))
    _L.call("hfmi_tuning_set", b"prof_level", timed_level)
    ctx.profile_begin()
    step_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        d, U = step()                                # returns when the eigenvalues are on the host: a complete solve
        step_ms.append((time.perf_counter() - ts) * 1e3)
    barrier()
    elapsed = time.perf_counter() - t0
    sysfs_after = gpu_sysfs_snapshot(pci) if rank == 0 else {}
    prof = ctx.profile_end()
    phases = ctx.profile_phases()
    phase_steps = args.steps
    if timed_level < 2:
        _L.call("hfmi_tuning_set", b"prof_level", 2)
        phase_steps = max(1, min(3, args.steps))
        ctx.profile_begin()
        for _ in range(phase_steps):
            step()
        barrier()
        ctx.profile_end()
        phases = ctx.profile_phases()
    median_ms = float(np.median(step_ms))
    if use_dist:
        elapsed = collective.allReduceMax(elapsed)   # the slowest rank's clock
        median_ms = collective.allReduceMax(median_ms)
    # the same solve with T = (A Q)^T Q formed literally, as the reference does (4 long contractions instead of 3):
    # reported next to the headline number, never part of it
    literal_ms = None
    if not args.no_literal and B is None:
        nl = max(1, min(3, args.steps))
        step(literal_T=True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(nl):
            step(literal_T=True)
        barrier()
        literal_ms = (time.perf_counter() - t0) / nl * 1e3
        if use_dist:
            literal_ms = collective.allReduceMax(literal_ms)
    comm_info = {"ranks": collective.size(), "transport": getattr(collective, "transport", "none"),
                 "launcher": os.environ.get("HFMI_LAUNCHER", "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "none")}
    if hasattr(collective, "describe"):
        comm_info.update(collective.describe())      # why this transport (RCCL first-contact fallback), PCI bus id of every rank
        comm_info["row_panels_overlapped"] = int(os.environ.get("HFMI_COMM_PANELS", "4"))
    comm_info["step_ms_max_over_ranks"] = elapsed / args.steps * 1e3
    if use_dist:
        collective.close()                           # collective: every rank leaves the communicator here;
    if rank != 0:                                    # rank 0 goes on alone with the oracle legs
        return None

    ms_per_step = elapsed / args.steps * 1e3
    value = N * r / (elapsed / args.steps) / 1e9
    out = {"metric": "randomized-SVD throughput (GDoF*rank/s)", "value": value, "unit": "GDoF*rank/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic (seeded, generated in HBM: SURVEY.md section 8d recipes)",
           "config": desc, "communicator": comm_info, "build_tag": hf.build_tag()}
    # SURVEY 8d defines the metric on the MEDIAN solve; `value` keeps the driver's contract (K steps between two barriers,
    # slowest rank), the median of the per-step clocks is reported beside it (a complete solve ends with its eigenvalues on
    # the host, so each step is bracketed by a synchronisation of its own)
    out["median_ms_per_step"] = median_ms
    out["value_from_median"] = N * r / (median_ms * 1e-3) / 1e9
    out["step_ms_min_max"] = [float(min(step_ms)), float(max(step_ms))]
    out["gpu_sysfs"] = {"pci_bus_id": pci, "before_timed_region": sysfs_before, "after_timed_region": sysfs_after}
    if literal_ms is not None:
        out["literal_T_ms_per_step"] = literal_ms
    out["phases_ms_per_step"] = {name: ms / phase_steps for name, ms in phases.items()}
    out["phases_from"] = ("the timed steps" if timed_level >= 2 else
                          "%d extra untimed steps with per-phase events (the timed steps carry events on the big contractions only)" % phase_steps)

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
        # correction + WRITE_SIZE; profiles/pmc_traffic.json) -- bench.py cannot run the profiler on itself.  The
        # record is used only if it was measured on THIS build of the kernels.
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            rec = tr["kernels"].get(out["roofline"]["kernel"])
            if rec and tr.get("build_tag") == hf.build_tag():
                out["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = tr["source"]
                out["roofline"]["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / pk["bytes_per_launch"]
                out["roofline"]["pmc_mfma_pipe_util"] = rec.get("mfma_pipe_util")
                out["roofline"]["pmc_effective_clock_ghz"] = rec.get("effective_clock_ghz")
            elif rec:
                out["roofline"]["traffic_note"] = ("profiles/pmc_traffic.json was measured on build %s, this is build %s: record not used"
                                                   % (tr.get("build_tag"), hf.build_tag()))
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
        if "roofline" in out:     # the same fraction against what this box sustains in this job (constant-operand MFMA loop / copy kernel)
            pm = out["device_peaks_measured"]
            out["roofline"]["frac_of_in_job_measured_peak"] = (out["roofline"]["fp64_mfma_frac"] * FP64_MFMA_PEAK_TFLOPS / pm["mfma_f64_tflops"]
                                                               if out["roofline"]["bound"] == "mfma" else
                                                               out["roofline"]["algorithmic_gbs"] / pm["hbm_copy_gbs"])
        # ... and against the MFMA rate the box sustains while HBM is being streamed (the regime the contractions run in)
        out["device_peaks_measured"].update(ctx.bench_loaded_peak())
        if "roofline" in out and out["roofline"]["bound"] == "mfma":
            out["roofline"]["frac_of_in_job_loaded_peak"] = (out["roofline"]["achieved"] /
                                                             out["device_peaks_measured"]["mfma_f64_tflops_while_streaming"])
        # ... and against the same loop on Gaussian operands (the constant-operand loops above toggle almost no bits: their clock
        # is one the solve's data never sees under the power limit)
        out["device_peaks_measured"].update(ctx.bench_hbm_read())
        out["device_peaks_measured"].update(ctx.bench_random_peaks())
        if "roofline" in out and out["roofline"]["bound"] == "mfma":
            pm = out["device_peaks_measured"]
            out["roofline"]["frac_of_in_job_random_operand_peak"] = out["roofline"]["achieved"] / pm["mfma_f64_tflops_random_operands"]
            out["roofline"]["frac_of_in_job_random_operand_peak_while_streaming"] = (
                out["roofline"]["achieved"] / pm["mfma_f64_tflops_random_operands_while_streaming"])
    except Exception as exc:   # the micro-benchmark is informative only
        out.setdefault("device_peaks_measured", {})["error"] = str(exc)

    from oracle import hippyflow_restated as hf_o
    from oracle import hippylib_restated as hp_o
    Omega_host = np.asfortranarray(Omega.to_dense())
    if not args.no_check:
        t0 = time.perf_counter()
        (d_ref, U_ref), apply_B = host_reference(args, hf, wl, prior, world, Omega_host, r, hp_o)
        Ud = np.asfortranarray(U.to_dense())
        lead = max(1, r // 2)
        angle = hp_o.principal_angle(Ud[:, :lead], U_ref[:, :lead], apply_B)
        BU = Ud if apply_B is None else apply_B(Ud)
        out["parity"] = {"eig_rel_err_vs_oracle": hp_o.eig_rel_err(d, d_ref), "principal_angle_rad_leading_%d" % lead: angle,
                         "orthonormality_defect": float(np.abs(Ud.T @ BU - np.eye(r)).max()),
                         "eigenvalue_range": [float(d[0]), float(d[-1])],
                         "relative_gap_below_leading_%d" % lead: float((d_ref[lead - 1] - d_ref[lead]) / d_ref[lead - 1]),
                         "oracle_seconds": time.perf_counter() - t0,
                         "oracle_form": "double_pass_blas3",
                         "note": "oracle = CPU restatement of the reference path (oracle/) on the same Omega and the same operator data, "
                                 "streamed from HBM to the host and applied densely there.  At this size the oracle leg is the BLAS-3 twin "
                                 "(oracle/hippylib_restated.py double_pass_blas3: block applies, Householder QR with positive diagonal, "
                                 "eigh) -- the thin QR with positive diagonal is unique, so Q, T, d and span(U) equal those of the "
                                 "column-by-column MGS restatement (double_pass / double_pass_g) to round-off; the column form itself is what "
                                 "the -m gpu tests and smoke() compare with at sizes it finishes in seconds.  Parity against hippylib "
                                 "itself is unpinned (hippylib is absent from the reference tree); LAPACK fixtures "
                                 "(tests/golden/independent_eig.npz) hold both the oracle and the device path to 1e-9"}
    if prior is not None and not args.quick:
        # feasibility estimate only (no kernel): what a level-scheduled device triangular solve would face in place of the host LU
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            from sptrsv_feasibility import estimate
            out["sptrsv_feasibility"] = estimate(prior.A, nrhs=r + p)
        except Exception as exc:
            out["sptrsv_feasibility"] = {"error": repr(exc)}
    if not args.no_cpu_baseline:
        # rank 0 alone (the other ranks have left the communicator): at N > 1 `wl` is rank 0's shard, the bounded sample is
        # drawn from it and scaled to the whole job exactly as at N = 1
        out["cpu_baseline"] = cpu_baseline(args, wl, prior, Omega_host, r, N, hp_o, hf_o)
    # release the workload's HBM before anything else is started on this GPU (the extras are child processes)
    del wl, op, A, Omega, U, B, Binv
    import gc
    gc.collect()
    ctx.synchronize()
    return out


if __name__ == "__main__":
    main()
