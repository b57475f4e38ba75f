"""Seeded synthetic inputs of the BASELINE configs, generated directly in HBM.

No datasets or PDE solvers exist in this environment, so each workload is built from a latent
factorisation whose operator spectrum is known in closed form:

* POD   (configs 1, 3):  snapshots X = U0 diag(sigma) W0^T with W0 (N x c) orthonormal in HBM (device QR of a
  Philox Gaussian block) and U0 (n x c) orthonormal (host QR) => (1/n) X^T X = W0 diag(sigma^2/n) W0^T.
* AS    (config 4):      J_i = A_i P^T, P (N x c) orthonormal, A_i = G_i diag(s) with G_i Gaussian keyed by the
  GLOBAL sample index (any rank regenerates exactly its own shard) => mean J^T J = P H P^T,
  H = mean A_i^T A_i (c x c).
* KLE   (config 2):      explicit dense covariance C = F diag(lam) F^T (N x N in HBM, 80 GB at N = 1e5) with
  F orthonormal, and the consistent P1 mass matrix of an nx x ny grid (CSR, 7 nnz/row).

The latent factors are kept so that tests and bench.py can evaluate the SAME operator on the host in
factored form (a few N x c products) -- that is what makes an oracle check affordable at full size.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .multivector import MultiVector, MvDSmatMult
from .operators import CsrOperator, MeanJTJfromDataOperator, SnapshotGramOperator, npToDeviceOperator


class Workload:
    pass


def _orthonormal_block(N, c, seed, stream, ctx):
    """N x c block with orthonormal columns: device QR of a Philox Gaussian draw (c <= 256)."""
    W = MultiVector(int(N), int(c), ctx=ctx)
    L.call("hfmi_randn_fill", W.handle, C.c_uint64(int(seed) & 0xFFFFFFFF), C.c_uint32(int(stream)), 1.0)
    W.orthogonalize(L.QR_CHOL)
    return W


def _expand(W0, S, ctx):
    """block (N x m) = W0 (N x c) * S (c x m), m arbitrary (panels of 256 inside the library)."""
    out = MultiVector(W0.size(), int(S.shape[1]), ctx=ctx)
    MvDSmatMult(W0, np.ascontiguousarray(S), out)
    return out


def pod_workload(N, n, latent=256, rate=0.05, seed=3, ctx=None, first_snapshot=0, n_local=None):
    """n snapshots of length N; exact eigenvalues of (1/n) X^T X are sigma_j^2 / n.  With ``n_local`` the workload holds
    only the snapshots first_snapshot ... first_snapshot + n_local - 1 of the SAME n-snapshot set (the per-rank shard of a
    snapshot-parallel run, PODProjector.py:359-363): its operator is the shard mean (1/n_local) X_g^T X_g, and the rank
    average of P equal shards is the full operator."""
    ctx = ctx or L.Context.default()
    c = int(min(latent, n, 256))
    n_local = int(n if n_local is None else n_local)
    wl = Workload()
    wl.N, wl.n, wl.latent, wl.n_local, wl.first_snapshot = int(N), int(n), c, n_local, int(first_snapshot)
    wl.W0 = _orthonormal_block(N, c, seed, 100, ctx)
    rng = np.random.default_rng(seed)
    U0, _ = np.linalg.qr(rng.standard_normal((n, c)))
    wl.sigma = np.exp(-rate * np.arange(c))
    wl.U0 = U0
    S = (U0 * wl.sigma).T                               # c x n
    wl.X = _expand(wl.W0, S[:, first_snapshot:first_snapshot + n_local], ctx)   # one snapshot per vector
    wl.operator = SnapshotGramOperator(wl.X, scale=1.0 / n_local)
    wl.exact_eigenvalues = wl.sigma ** 2 / n
    return wl


def pod_host_apply(wl, W0_host):
    """Same operator on the host in factored form: W -> W0 diag(sigma^2/n) (W0^T W) up to the (exactly
    orthonormal in exact arithmetic) U0 factor, which is kept: (1/n) W0 S S^T W0^T."""
    S = (wl.U0 * wl.sigma).T
    H = S @ S.T / wl.n
    return lambda W: np.asfortranarray(W0_host @ (H @ (W0_host.T @ W)))


NOISE_STREAM0 = 0x10000     # Philox stream of sample i's noise block E_i: NOISE_STREAM0 + global sample index


def as_workload(N, ns_local, q=100, latent=100, rate=0.06, seed=4, first_sample=0, ns_total=None, ctx=None,
                noise_cov_inv=None, noise=0.0, P=None):
    """ns_local Jacobian samples (global indices first_sample ...) of shape q x N, stored as one block of
    ns_local*q vectors.  ``scale`` of the returned operator is 1/ns_local (the per-rank mean); the rank
    average is the collective's job.

    ``noise``: SURVEY section 8d's config 4 is ``J_i = A_i P^T + 0.01 E_i`` with E_i (q x N) i.i.d. Gaussian keyed
    by the GLOBAL sample index (device Philox: seed, stream NOISE_STREAM0 + i, vector = row o, the element map of
    ``hfmi_randn_fill``), so any rank regenerates exactly its shard and a host can regenerate E_i bit-for-bit from the
    same counters.  With noise the operator is full rank (no closed-form spectrum): parity is then checked against a
    dense host evaluation of the same J."""
    ctx = ctx or L.Context.default()
    c = int(latent)
    wl = Workload()
    wl.N, wl.ns_local, wl.q, wl.latent = int(N), int(ns_local), int(q), c
    wl.first_sample = int(first_sample)
    wl.ns_total = int(ns_total if ns_total is not None else ns_local)
    wl.P = P if P is not None else _orthonormal_block(N, c, seed, 200, ctx)     # shared by all samples (and ranks)
    wl.s = np.exp(-rate * np.arange(c))
    wl.A = np.stack([sample_factor(seed, first_sample + i, q, c) * wl.s[None, :] for i in range(ns_local)])   # (ns, q, c)
    S = wl.A.reshape(ns_local * q, c).T                 # c x (ns*q): column i*q+o = row o of A_i
    wl.J = _expand(wl.P, S, ctx)
    wl.noise, wl.seed = float(noise), int(seed)
    if noise:
        E = MultiVector(int(N), int(q), ctx=ctx)
        for i in range(ns_local):             # Philox key = seed, stream = NOISE_STREAM0 + global sample index (hfmi_randn_fill)
            L.call("hfmi_randn_fill", E.handle, C.c_uint64(int(seed) & 0xFFFFFFFF), C.c_uint32(NOISE_STREAM0 + first_sample + i), 1.0)
            wl.J.view(i * q, q).axpy(float(noise), E)
    wl.operator = MeanJTJfromDataOperator.from_block(wl.J, ns_local, q, noise_cov_inv=noise_cov_inv)
    return wl


def sample_factor(seed, global_index, q, c):
    """Gaussian q x c factor of sample ``global_index`` (numpy Philox keyed by (seed, index))."""
    bitgen = np.random.Philox(key=[int(seed), int(global_index)])
    return np.random.Generator(bitgen).standard_normal((q, c))


def as_reduced_matrix(seed, ns_total, q, c, rate):
    """H = (1/ns_total) sum_i A_i^T A_i over ALL samples: mean J^T J = P H P^T."""
    s = np.exp(-rate * np.arange(c))
    H = np.zeros((c, c))
    for i in range(ns_total):
        A = sample_factor(seed, i, q, c) * s[None, :]
        H += A.T @ A
    return H / ns_total


def grid_mass_matrix(nx, ny):
    """Consistent P1 mass matrix of an nx x ny structured triangulation of the unit square (CSR, <= 7 nnz/row)."""
    import scipy.sparse as sp
    hx, hy = 1.0 / (nx - 1), 1.0 / (ny - 1)
    area = 0.5 * hx * hy
    idx = np.arange(nx * ny).reshape(ny, nx)
    v00, v10, v01, v11 = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    tris = np.concatenate([np.stack([v00, v10, v11], 1), np.stack([v00, v11, v01], 1)])
    rows, cols, vals = [], [], []
    for a in range(3):
        for b in range(3):
            rows.append(tris[:, a])
            cols.append(tris[:, b])
            vals.append(np.full(len(tris), area / 6.0 if a == b else area / 12.0))
    M = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nx * ny, nx * ny))
    return M.tocsr()


def grid_stiffness_matrix(nx, ny):
    """P1 stiffness matrix (Neumann) of the same nx x ny triangulation of the unit square as ``grid_mass_matrix``."""
    import scipy.sparse as sp
    hx, hy = 1.0 / (nx - 1), 1.0 / (ny - 1)
    idx = np.arange(nx * ny).reshape(ny, nx)
    v00, v10, v01, v11 = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    rx, ry = 0.5 * hy / hx, 0.5 * hx / hy
    rows, cols, vals = [], [], []

    def add(a, b, v):
        rows.append(a)
        cols.append(b)
        vals.append(np.full(len(a), v))

    # both triangles of a cell are right-angled: (a, b, c) with the right angle at b, a-b along x, b-c along y
    for a, b, c in ((v00, v10, v11), (v11, v01, v00)):
        add(a, a, rx), add(b, b, rx + ry), add(c, c, ry)
        add(a, b, -rx), add(b, a, -rx), add(b, c, -ry), add(c, b, -ry)
    K = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nx * ny, nx * ny))
    return K.tocsr()


def _lu_worker(conn, csc_parts, ml, x_name, y_name, N, kmax):
    """One worker PROCESS of SparseLUPriorSolver: its own SuperLU factorisation of A, slabs through shared memory.
    Started with the 'spawn' method (a fresh interpreter that never touches the GPU; nothing is forked from, or exec'ed
    over, a process that has initialised HIP)."""
    import numpy as _np
    import scipy.sparse as _sp
    import scipy.sparse.linalg as _spla
    from multiprocessing import shared_memory as _shm
    try:
        from threadpoolctl import threadpool_limits as _limits
        _ctx = _limits(limits=1)
    except ImportError:
        import contextlib
        _ctx = contextlib.nullcontext()
    data, indices, indptr = csc_parts
    with _ctx:
        lu = _spla.splu(_sp.csc_matrix((data, indices, indptr), shape=(N, N)))
        xs, ys = _shm.SharedMemory(name=x_name), _shm.SharedMemory(name=y_name)
        X = _np.ndarray((kmax, N), dtype=_np.float64, buffer=xs.buf)
        Y = _np.ndarray((kmax, N), dtype=_np.float64, buffer=ys.buf)
        conn.send("ready")
        while True:
            msg = conn.recv()
            if msg is None:
                break
            r0, r1 = msg
            try:
                Z = lu.solve(X[r0:r1].T)             # (N, cnt) Fortran view of the slab's rows: no copy
                Z *= ml[:, None]
                Y[r0:r1] = lu.solve(_np.asfortranarray(Z)).T
                conn.send(("ok", r0, r1))
            except Exception as exc:                 # noqa: BLE001 -- reported to the parent, which raises
                conn.send(("error", repr(exc)))
        del X, Y
        xs.close()
        ys.close()
    conn.close()


class SparseLUPriorSolver:
    """``prior.Rsolver`` of a bi-Laplacian prior, R^-1 = A^-1 M_l A^-1, as a HOST black box: sparse LU of A (SuperLU)
    and two triangular-solve sweeps per vector.  ``solve(y, x)`` is the reference's solver protocol on 1-D arrays;
    ``solve_block(X)`` serves (N, k) slabs.

    ``processes > 1``: the vectors of a slab are dealt to a pool of worker processes, each holding its OWN factorisation;
    the slabs travel through two shared-memory segments (vector-major, so a worker's share is a Fortran view, no copy).
    SuperLU's triangular sweeps hold the GIL and do not scale over Python threads (74 vectors: 1.6 s on 1 thread, 6.3 s on
    32, scripts/host_solver_probe.py) -- processes do.  The reference is in the same situation with one PETSc solve per
    vector per MPI rank (activeSubspaceProjector.py:447-453)."""

    def __init__(self, A, M_lumped, threads=None, processes=None):
        import scipy.sparse.linalg as spla
        self.N = A.shape[0]
        self._A = A.tocsc()
        self.Ml = np.asarray(M_lumped, dtype=np.float64)
        self.threads = int(threads or 1)
        self.processes = int(processes or 1)
        self.lu = spla.splu(self._A) if self.processes <= 1 else None
        self._pool = None
        self._workers = None
        self.calls, self.vectors = 0, 0

    # ---- process pool -------------------------------------------------------------------
    def _start_workers(self, kmax):
        import multiprocessing as mp
        from multiprocessing import shared_memory
        self.close()
        ctx = mp.get_context("spawn")
        nbytes = max(8, kmax * self.N * 8)
        self._xs = shared_memory.SharedMemory(create=True, size=nbytes)
        self._ys = shared_memory.SharedMemory(create=True, size=nbytes)
        self._X = np.ndarray((kmax, self.N), dtype=np.float64, buffer=self._xs.buf)
        self._Y = np.ndarray((kmax, self.N), dtype=np.float64, buffer=self._ys.buf)
        self._kmax = kmax
        parts = (self._A.data, self._A.indices, self._A.indptr)
        self._workers = []
        for _ in range(self.processes):
            parent, child = ctx.Pipe()
            pr = ctx.Process(target=_lu_worker, args=(child, parts, self.Ml, self._xs.name, self._ys.name, self.N, kmax), daemon=True)
            pr.start()
            child.close()
            self._workers.append((pr, parent))
        for pr, conn in self._workers:
            if conn.recv() != "ready":
                raise RuntimeError("SparseLUPriorSolver: a worker process did not start")

    def close(self):
        """Stop the worker processes and remove the shared-memory segments (they live in /dev/shm)."""
        if self._workers:
            for pr, conn in self._workers:
                try:
                    conn.send(None)
                except (OSError, BrokenPipeError):
                    pass
            for pr, conn in self._workers:
                pr.join(timeout=10)
                if pr.is_alive():
                    pr.terminate()          # exactly the processes started here
                conn.close()
            self._workers = None
            self._X = self._Y = None
            for seg in (self._xs, self._ys):
                seg.close()
                seg.unlink()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _solve_in_pool(self, X):
        N, k = X.shape
        if self._workers is None or k > self._kmax:
            self._start_workers(max(k, 32))
        self._X[:k] = X.T                            # vector-major slab
        step = max(1, -(-k // self.processes))
        busy = []
        for w, r0 in enumerate(range(0, k, step)):
            conn = self._workers[w][1]
            conn.send((r0, min(k, r0 + step)))
            busy.append(conn)
        for conn in busy:
            msg = conn.recv()
            if msg[0] != "ok":
                raise RuntimeError("SparseLUPriorSolver worker: %s" % msg[1])
        return self._Y[:k].T.copy(order="F")

    # ---- in-process -----------------------------------------------------------------------
    def _one(self, X):
        Z = self.lu.solve(np.asfortranarray(X))
        Z *= self.Ml[:, None]
        return self.lu.solve(Z)

    def _blas_serial(self):
        """SuperLU's triangular sweeps call level-2 BLAS on small supernodes: a threaded BLAS only adds overhead there
        (on a 256-core host it made the solve 4x slower).  The parallelism here is over vectors."""
        try:
            from threadpoolctl import threadpool_limits
            return threadpool_limits(limits=1)
        except ImportError:
            import contextlib
            return contextlib.nullcontext()

    def solve_block(self, X):
        X = np.asarray(X)
        k = X.shape[1]
        self.calls, self.vectors = self.calls + 1, self.vectors + k
        if self.processes > 1:
            return self._solve_in_pool(X)
        with self._blas_serial():
            if self.threads <= 1 or k == 1:
                return self._one(X)
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(self.threads)
            step = max(1, -(-k // self.threads))
            parts = list(self._pool.map(lambda c: self._one(X[:, c:c + step]), range(0, k, step)))
        return np.concatenate(parts, axis=1)

    def solve(self, y, x):
        if self.processes > 1:
            y[...] = self._solve_in_pool(np.asarray(x).reshape(-1, 1))[:, 0]
            return
        with self._blas_serial():
            y[...] = self._one(np.asarray(x).reshape(-1, 1))[:, 0]


class BiLaplacianPrior:
    """The prior shape of the reference's tests (hp.BiLaplacianPrior, test_derivativeSubspace.py:40):
    precision R = A M_l^-1 A with A = delta M + gamma K on an nx x ny P1 grid, ``R`` as a sparse matrix (applied on the
    device as CSR), ``Rsolver`` a host sparse-LU black box, ``M`` the consistent mass matrix."""

    def __init__(self, nx, ny, delta=1.0, gamma=0.1, threads=None, processes=None):
        import scipy.sparse as sp
        self.nx, self.ny, self.delta, self.gamma = nx, ny, delta, gamma
        self.M = grid_mass_matrix(nx, ny)
        self.K = grid_stiffness_matrix(nx, ny)
        self.A = (delta * self.M + gamma * self.K).tocsr()
        self.M_lumped = np.asarray(self.M.sum(axis=1)).ravel()
        self.R = (self.A @ sp.diags(1.0 / self.M_lumped) @ self.A).tocsr()
        self.Rsolver = SparseLUPriorSolver(self.A, self.M_lumped, threads=threads, processes=processes)

    def init_vector(self, x, dim):
        x.init(self.R.shape[0])


def matern32_covariance(C, nx, ny, sigma=1.0, ell=0.1):
    """Fill the N x N block C (N = C.size() <= nx*ny grid nodes, row-major node numbering on the unit square) with the
    Matern-3/2 kernel sigma^2 (1 + sqrt(3) d / ell) exp(-sqrt(3) d / ell) of SURVEY section 8d's config 2, on the device."""
    L.call("hfmi_block_fill_matern32", C.handle, int(nx), int(ny), float(sigma), float(ell))
    return C


def matern32_host(N, nx, ny, sigma=1.0, ell=0.1, rows=None):
    """The same kernel on the host (rows ``rows`` of the N x N matrix; all rows by default): the dense evaluation the
    miniature fixture and the streaming oracle use."""
    idx = np.arange(N)
    x, y = (idx % nx) / (nx - 1.0), (idx // nx) / (ny - 1.0)
    r = idx if rows is None else np.asarray(rows)
    d = np.sqrt((x[r, None] - x[None, :]) ** 2 + (y[r, None] - y[None, :]) ** 2)
    a = np.sqrt(3.0) * d / ell
    return sigma ** 2 * (1.0 + a) * np.exp(-a)


def kle_matern_workload(nx, ny, N=None, sigma=1.0, ell=0.1, ctx=None):
    """SURVEY 8d config 2 as specified: explicit dense Matern-3/2 covariance on the first N nodes of an nx x ny grid
    (N = 1e5 of 316 x 317), the P1 mass matrix of that grid restricted to the same nodes."""
    ctx = ctx or L.Context.default()
    N = int(nx * ny if N is None else N)
    wl = Workload()
    wl.N, wl.nx, wl.ny, wl.sigma, wl.ell = N, nx, ny, sigma, ell
    wl.C = MultiVector(N, N, ctx=ctx)
    matern32_covariance(wl.C, nx, ny, sigma, ell)
    wl.M = grid_mass_matrix(nx, ny)[:N, :N].tocsr()
    wl.C_operator = npToDeviceOperator(wl.C)
    wl.M_operator = CsrOperator(wl.M, ctx=ctx)
    return wl


def kle_workload(nx, ny, latent=256, rate=0.08, seed=2, ctx=None):
    """Explicit dense covariance C = F diag(lam) F^T on N = nx*ny points and the grid mass matrix."""
    ctx = ctx or L.Context.default()
    N = nx * ny
    c = int(min(latent, 256))
    wl = Workload()
    wl.N, wl.latent = N, c
    wl.F = _orthonormal_block(N, c, seed, 300, ctx)
    wl.lam = np.exp(-rate * np.arange(c))
    Fh = wl.F.to_dense()                                # N x c on the host (for the factored oracle as well)
    wl.F_host = Fh
    S = (Fh * wl.lam).T                                 # c x N
    wl.C = _expand(wl.F, S, ctx)                        # N x N symmetric, column j = C[:, j]
    wl.M = grid_mass_matrix(nx, ny)
    wl.C_operator = npToDeviceOperator(wl.C)
    wl.M_operator = CsrOperator(wl.M, ctx=ctx)
    return wl


class _DipnetObservable:
    """The observable hooks the projectors ask for (projectors.py docstring), backed by the synthetic map."""

    def __init__(self, wl):
        self.wl = wl

    def jacobian_data(self, n):
        assert n == self.wl.ns
        return self.wl.J, self.wl.ns, self.wl.dQ

    def input_dimension(self):
        return self.wl.dM

    def output_dimension(self):
        return self.wl.dQ

    def sample_observables(self, n, prior, noise):
        return self.wl.q_train[:n].astype(np.float64)


def dipnet_workload(dM=20000, dQ=400, hidden=80, n_train=8192, n_test=1024, ns=64, rate=0.08, seed=5, ctx=None):
    """BASELINE config 5's data source in synthetic form: a smooth nonlinear parameter-to-observable map
    q(m) = W2 tanh(W1^T m) with W1 = P diag(sigma) (P: dM x hidden orthonormal in HBM, sigma_j = exp(-rate j)) and
    Gaussian W2, inputs m ~ N(0, I).  Its Jacobian at m_i is J_i = W2 diag(1 - tanh^2(W1^T m_i)) W1^T, so the stacked
    Jacobians of ``ns`` training points are one expansion of the block P -- the same layout the AS projector consumes
    (the reference obtains them from PDE solves, activeSubspaceProjector.py:347-397)."""
    ctx = ctx or L.Context.default()
    rng = np.random.default_rng(seed)
    wl = Workload()
    wl.dM, wl.dQ, wl.hidden, wl.ns = int(dM), int(dQ), int(hidden), int(ns)
    wl.P = _orthonormal_block(dM, hidden, seed, 400, ctx)
    wl.sigma = np.exp(-rate * np.arange(hidden))
    Ph = wl.P.to_dense()                                             # dM x hidden
    wl.W1 = (Ph * wl.sigma).astype(np.float32)
    wl.W2 = (rng.standard_normal((dQ, hidden)) / np.sqrt(hidden)).astype(np.float32)
    m = rng.standard_normal((n_train + n_test, dM)).astype(np.float32)
    z = m @ wl.W1
    q = np.tanh(z) @ wl.W2.T
    wl.m_train, wl.q_train, wl.m_test, wl.q_test = m[:n_train], q[:n_train], m[n_train:], q[n_train:]
    # J_i = (W2 D_i diag(sigma)) P^T: row o of sample i is vector i*dQ + o of the block
    D = 1.0 - np.tanh(z[:ns].astype(np.float64)) ** 2                # ns x hidden
    S = (wl.W2.astype(np.float64)[None, :, :] * (D * wl.sigma)[:, None, :]).reshape(ns * dQ, hidden).T   # hidden x (ns*dQ)
    wl.J = _expand(wl.P, np.ascontiguousarray(S), ctx)
    wl.observable = _DipnetObservable(wl)
    return wl
