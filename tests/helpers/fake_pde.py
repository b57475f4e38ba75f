"""A small nonlinear 'PDE' in numpy with the object layout of a hIPPYlib problem, for exercising the reference's
observable / prior / vector protocols without FEniCS.

Used twice, with the SAME numbers:
* ``tests/golden/make_goldens.py`` wraps ``NumpyProblem`` + ``ObservationOperator`` in the REFERENCE's own
  ``LinearStateObservable`` and runs the reference's projector code over it (fixtures ``protocol_*.npz``);
* the GPU tests wrap them in ``ProtocolObservable`` below -- a class exposing exactly the method names of the reference's
  observable (modeling/observable.py:66-323) and nothing else -- and hand it to ``hippyflow_amd``'s projectors.

State equation:  (K + diag(exp(m))) u = f  with a non-symmetric K (so that forward and adjoint incremental solves
differ).  Linearised at (u, m):  A = K + diag(exp(m)),  C = d(residual)/dm = diag(exp(m) * u),  Jacobian of the
observable q = B u:  J = -B A^{-1} C.

Every method takes and fills dolfin-like vectors (``get_local`` / ``set_local`` / ``init`` / ``zero`` / ``axpy``); which
vector class is used is the caller's business (the generator's stand-in ``dolfin.Vector``, ``hippyflow_amd.HostVector``).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

STATE, PARAMETER, ADJOINT, CONTROL = 0, 1, 2, 3


class _Comm:
    rank = 0
    size = 1

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1


class _Mesh:
    def mpi_comm(self):
        return _Comm()


class _Space:
    def mesh(self):
        return _Mesh()


class MatrixOperator:
    """A (possibly rectangular) matrix behind ``mult / transpmult / init_vector(x, dim) / mpi_comm`` -- the surface of a
    dolfin matrix that the reference touches."""

    def __init__(self, A, export_csr=False):
        self.A = sp.csr_matrix(A) if sp.issparse(A) else np.asarray(A)
        self.calls = 0
        if export_csr:                          # what a PETSc matrix offers (PODProjector.py:322-324)
            self.getValuesCSR = self._get_values_csr

    def mpi_comm(self):
        return _Comm()

    def init_vector(self, x, dim):
        x.init(self.A.shape[0] if dim == 0 else self.A.shape[1])

    def mult(self, x, y):
        self.calls += 1
        y.set_local(self.A @ x.get_local())

    def transpmult(self, x, y):
        self.calls += 1
        y.set_local(self.A.T @ x.get_local())

    def _get_values_csr(self):
        M = sp.csr_matrix(self.A)
        return M.indptr, M.indices, M.data

    def getSize(self):
        return self.A.shape


class FactorizedSolver:
    """``solve(x, b)`` (and ``init_vector``) over a sparse LU: the surface of a PETSc solver / hippylib's Rsolver."""

    def __init__(self, A, with_init_vector=True):
        self.n = A.shape[0]
        self.lu = spla.splu(sp.csc_matrix(A))
        self.calls = 0
        if with_init_vector:
            self.init_vector = self._init_vector

    def _init_vector(self, x, dim):
        x.init(self.n)

    def solve(self, x, b):
        self.calls += 1
        x.set_local(self.lu.solve(b.get_local()))


class NumpyProblem:
    """The 'PDEProblem': forward solve, linearisation point, incremental solves and the C block."""

    def __init__(self, n, vector_class, seed=0):
        rng = np.random.default_rng(seed)
        self.n = n
        self._vec = vector_class
        h = 1.0 / (n + 1)
        diff = sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]) / h ** 2 * 1e-2
        conv = sp.diags([-np.ones(n - 1), np.ones(n - 1)], [-1, 1]) / (2 * h) * 0.05         # makes K non-symmetric
        self.K = sp.csr_matrix(diff + conv)
        self.f = 1.0 + 0.5 * np.sin(np.linspace(0, 3 * np.pi, n)) + 0.1 * rng.standard_normal(n)
        self.Vh = [_Space(), _Space(), _Space()]
        self.C = MatrixOperator(sp.identity(n))      # not None: "the KKT blocks exist" (activeSubspaceProjector.py:523)
        self.fail_every = 0                          # > 0: every fail_every-th forward solve raises
        self.n_fwd = self.n_inc = 0
        self._A = None
        self._Cdiag = None

    def _new(self):
        v = self._vec()
        v.init(self.n)
        return v

    def generate_state(self):
        return self._new()

    def generate_parameter(self):
        return self._new()

    def init_parameter(self, m):
        m.init(self.n)

    def _operator(self, m):
        return sp.csc_matrix(self.K + sp.diags(np.exp(m)))

    def solveFwd(self, out, x):
        self.n_fwd += 1
        if self.fail_every and self.n_fwd % self.fail_every == 0:
            raise RuntimeError("Newton did not converge")
        out.set_local(spla.spsolve(self._operator(x[PARAMETER].get_local()), self.forcing(x)))

    def setLinearizationPoint(self, x, gauss_newton_approx):
        u, m = x[STATE].get_local(), x[PARAMETER].get_local()
        self._A = spla.splu(self._operator(m))
        self._Cdiag = np.exp(m) * u

    def solveIncremental(self, sol, rhs, is_adjoint):
        self.n_inc += 1
        sol.set_local(self._A.solve(rhs.get_local(), trans='T' if is_adjoint else 'N'))

    def apply_ij(self, i, j, direction, out):
        assert (i, j) in ((ADJOINT, PARAMETER), (PARAMETER, ADJOINT))
        out.set_local(self._Cdiag * direction.get_local())          # C is diagonal: C and C^T coincide

    def forcing(self, x):
        return self.f

    def jacobian_dense(self, B):
        """-B A^{-1} C at the current linearisation point (for the checks, never used by a product path)."""
        AinvC = self._A.solve(np.diag(self._Cdiag))
        return -(B @ AinvC)


class NumpyControlProblem(NumpyProblem):
    """The same state equation with a control on the right-hand side:  (K + diag(exp(m))) u = f + G z,  z of length dz.
    d(residual)/dz = Cz = -G, so the Jacobian with respect to the control is  Jz = -B A^{-1} Cz = B A^{-1} G.  Having ``Cz`` is
    what makes a problem a control problem for the reference's observable (observable.py:79)."""

    def __init__(self, n, dz, vector_class, seed=0):
        super().__init__(n, vector_class, seed)
        self.dz = dz
        centres = (np.arange(dz) + 0.5) * n / dz
        self.G = np.exp(-0.5 * ((np.arange(n)[:, None] - centres[None, :]) / 3.0) ** 2)
        self.Cz = MatrixOperator(-self.G)

    def generate_control(self):
        v = self._vec()
        v.init(self.dz)
        return v

    def forcing(self, x):
        return self.f + self.G @ x[CONTROL].get_local()

    def apply_ij(self, i, j, direction, out):
        if (i, j) == (ADJOINT, CONTROL):
            out.set_local(-self.G @ direction.get_local())
        elif (i, j) == (CONTROL, ADJOINT):
            out.set_local(-self.G.T @ direction.get_local())
        else:
            super().apply_ij(i, j, direction, out)

    def control_jacobian_dense(self, B):
        return B @ self._A.solve(self.G)


class ControlDistribution:
    """What the projectors ask of a control distribution: ``sample(z)`` (and optionally ``mean``)."""

    def __init__(self, dz, seed=3):
        self.rng = np.random.default_rng(seed)
        self.dz = dz

    def sample(self, z):
        z.set_local(0.3 * self.rng.standard_normal(self.dz))


def observation_matrix(q, n, seed=1):
    rng = np.random.default_rng(seed)
    B = np.zeros((q, n))
    for i in range(q):
        c = (i + 0.5) * n / q
        B[i] = np.exp(-0.5 * ((np.arange(n) - c) / 2.0) ** 2)
    return B + 0.01 * rng.standard_normal((q, n))


class NumpyPrior:
    """BiLaplacian-shaped Gaussian prior: A = delta M + gamma K, R = A M^-1 A, sample = mean + A^-1 M^(1/2) noise."""

    def __init__(self, n, vector_class, delta=1.0, gamma=0.02, export_csr=False, mass_solver_shapes_vectors=True):
        self.n = n
        self._vec = vector_class
        h = 1.0 / (n + 1)
        main = np.full(n, 4.0 * h / 6.0)
        off = np.full(n - 1, h / 6.0)
        Mc = sp.diags([off, main, off], [-1, 0, 1], format="csr")            # consistent mass matrix (prior.M)
        Ml = np.asarray(Mc.sum(axis=1)).ravel()                              # lumped, inside R
        K = sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]) / h
        A = sp.csr_matrix(delta * sp.diags(Ml) + gamma * K)
        self.Asp, self.Ml = A, Ml
        self._Alu = spla.splu(sp.csc_matrix(A))
        self.Rmat = sp.csr_matrix(A @ sp.diags(1.0 / Ml) @ A)
        self.R = MatrixOperator(self.Rmat)
        self.Rsolver = FactorizedSolver(self.Rmat)
        self.M = MatrixOperator(Mc, export_csr=export_csr)
        self.Msolver = FactorizedSolver(Mc, with_init_vector=mass_solver_shapes_vectors)
        self.Vh = _Space()
        self.mean = vector_class()
        self.mean.init(n)
        self.mean.set_local(0.2 * np.cos(np.linspace(0, np.pi, n)))
        self.n_samples = 0

    def init_vector(self, x, dim):
        x.init(self.n)                       # "noise", 0 and 1 all have length n here

    def sample(self, noise, s, add_mean=True):
        self.n_samples += 1
        draw = self._Alu.solve(np.sqrt(self.Ml) * noise.get_local())
        s.set_local(draw + (self.mean.get_local() if add_mean else 0.0))


class ProtocolObservable:
    """The reference's observable surface (modeling/observable.py:66-323 ``LinearStateObservable``), method for method,
    over a problem + observation operator -- what an existing hippyflow driver hands to the projectors."""

    def __init__(self, problem, B):
        self.problem = problem
        self.B = B
        self.n_fwd_solve = self.n_adj_solve = self.n_inc_solve = 0
        self.is_control_problem = hasattr(problem, "Cz")
        if self.is_control_problem:                   # (observable.py:299-323)
            self.applyCz = lambda dz, out: self.problem.apply_ij(ADJOINT, CONTROL, dz, out)
            self.applyCzt = lambda dp, out: self.problem.apply_ij(CONTROL, ADJOINT, dp, out)

    def mpi_comm(self):
        return self.B.mpi_comm()

    def generate_vector(self, component="ALL"):
        if component == "ALL":
            x = [self.problem.generate_state(), self.problem.generate_parameter(), self.problem.generate_state()]
            return x + [self.problem.generate_control()] if self.is_control_problem else x
        if component == CONTROL:
            assert self.is_control_problem, 'Assuming it is a control problem'
            return self.problem.generate_control()
        return self.problem.generate_parameter() if component == PARAMETER else self.problem.generate_state()

    def init_vector(self, x, dim):
        if dim == 0:
            self.B.init_vector(x, 0)
        elif dim == 1:
            self.problem.C.init_vector(x, 1)
        elif dim == 3:
            assert self.is_control_problem, 'Assuming it is a control problem'
            self.problem.Cz.init_vector(x, 1)
        else:
            raise ValueError(dim)

    def evalu(self, u):
        out = type(u)()
        self.B.init_vector(out, 0)
        self.B.mult(u, out)
        return out

    def solveFwd(self, out, x):
        self.n_fwd_solve += 1
        self.problem.solveFwd(out, x)

    def setLinearizationPoint(self, x):
        x[ADJOINT] = self.problem.generate_state()
        self.problem.setLinearizationPoint(x, True)

    def solveFwdIncremental(self, sol, rhs):
        self.n_inc_solve += 1
        self.problem.solveIncremental(sol, rhs, False)

    def solveAdjIncremental(self, sol, rhs):
        self.n_inc_solve += 1
        self.problem.solveIncremental(sol, rhs, True)

    def applyB(self, x, out):
        self.B.mult(x, out)

    def applyBt(self, x, out):
        self.B.transpmult(x, out)

    def applyC(self, dm, out):
        self.problem.apply_ij(ADJOINT, PARAMETER, dm, out)

    def applyCt(self, dp, out):
        self.problem.apply_ij(PARAMETER, ADJOINT, dp, out)


SIZES = dict(n=48, q=7, n_samples=5, rank=6, oversampling=4, seed=1)
