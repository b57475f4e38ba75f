"""Fixtures from INDEPENDENT dense solvers (LAPACK through scipy), made with numpy/scipy only:

    python tests/golden/make_independent_goldens.py        ->  tests/golden/independent_eig.npz

They pin the unpinned part of the oracle (oracle/hippylib_restated.py restates hippylib, which is absent from
/root/reference) from a second direction: both the oracle and the HIP path are asserted against exact dense
eigen-decompositions at 1e-9.  With s = 3 power iterations and the spectra below, the randomisation error of the
leading r eigenvalues is far below that tolerance, so an error in the restated algorithm that the oracle and the
kernels shared would show.

  hep_*    A (N x N SPD, decaying spectrum), Omega, r, s; exact: scipy.linalg.eigh(A)
  ghep_*   the same A with an SPD tridiagonal B (P1 mass-matrix shape); exact: scipy.linalg.eigh(A, B)
  matern_* SURVEY.md section 8d's config-2 recipe in miniature: Matern-3/2 covariance (sigma=1, ell=0.1) on the first
           4000 nodes of a 64 x 63 grid, P1 mass matrix of that grid; exact: leading eigenvalues of
           scipy.linalg.eigh(M C M, M).  Only parameters and eigenvalues are stored (the matrices are regenerated).
"""
import os
import sys

import numpy as np
import scipy.linalg as sla

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    out = {}
    rng = np.random.default_rng(20261002)
    N, c, k, r, s = 300, 60, 20, 6, 3
    W, _ = np.linalg.qr(rng.standard_normal((N, N)))
    lam = np.concatenate([np.exp(-0.4 * np.arange(c)), 1e-9 * rng.random(N - c)])
    A = (W * lam) @ W.T
    A = 0.5 * (A + A.T)
    Omega = rng.standard_normal((N, k))
    d, V = sla.eigh(A)
    out.update(hep_A=A, hep_Omega=Omega, hep_r=r, hep_s=s, hep_d_exact=d[::-1][:r].copy(), hep_U_exact=V[:, ::-1][:, :r].copy())
    h = 1.0 / (N - 1)
    main_d = np.full(N, 4 * h / 6)
    main_d[[0, -1]] = 2 * h / 6
    B = np.diag(main_d) + np.diag(np.full(N - 1, h / 6), 1) + np.diag(np.full(N - 1, h / 6), -1)
    dg, Vg = sla.eigh(A, B)
    out.update(ghep_B_diag=main_d, ghep_B_off=np.full(N - 1, h / 6), ghep_d_exact=dg[::-1][:r].copy(), ghep_U_exact=Vg[:, ::-1][:, :r].copy())

    from hippyflow_amd import workloads                      # host-only helpers (scipy / numpy)
    nx, ny, Nm = 64, 63, 4000
    C = workloads.matern32_host(Nm, nx, ny, 1.0, 0.1)
    M = workloads.grid_mass_matrix(nx, ny)[:Nm, :Nm].toarray()
    MCM = M @ C @ M
    dm = sla.eigh(0.5 * (MCM + MCM.T), M, eigvals_only=True)
    out.update(matern_nx=nx, matern_ny=ny, matern_N=Nm, matern_sigma=1.0, matern_ell=0.1, matern_d_exact=dm[::-1][:40].copy(),
               matern_C_corner=C[:5, :5].copy(), matern_C_checksum=float(C.sum()))

    # the oracle must already agree with the exact solvers (checked here so that a bad fixture is never written)
    from oracle import hippylib_restated as hp_o
    d_o, _ = hp_o.double_pass(hp_o.DenseOperator(A), np.asfortranarray(Omega), r, s=s)
    assert np.max(np.abs(d_o - out["hep_d_exact"]) / out["hep_d_exact"]) < 1e-10, d_o

    class Solve:
        def solve(self, y, x):
            y[...] = np.linalg.solve(B, x)
    d_og, _ = hp_o.double_pass_g(hp_o.DenseOperator(A), hp_o.DenseOperator(B), Solve(), np.asfortranarray(Omega), r, s=s)
    assert np.max(np.abs(d_og - out["ghep_d_exact"]) / out["ghep_d_exact"]) < 1e-10, d_og
    np.savez_compressed(os.path.join(HERE, "independent_eig.npz"), **out)
    print("wrote independent_eig.npz:", {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
