"""Mixed-precision correction path: the block sweeps inside the GJD correction solve read an fp32 copy of
the stored symmetric tiles; residuals, projections and the convergence test stay on the fp64 matrix.  Opt-in
(engine_set_inner_precision / DAVIDSON_INNER_PRECISION=32).  Parity: the reference's golden GJD cases keep their
eigenvalues (1e-8), residuals (< tol) and outer iteration counts."""
import numpy as np
import pytest

import fortran_davidson_amd as fd
from fortran_davidson_amd.engine_c import OP_A, PANEL_V, PANEL_W, PANEL_R, METHOD_GJD
from oracle import davidson_oracle as O
from conftest import case_matrices
from test_solver_gpu import GJD_CASES, residuals, EV_TOL

pytestmark = pytest.mark.gpu


def test_inner_sweeps_on_fp32_tiles_solve_the_projected_systems_to_fp32_accuracy():
    """K7 with the fp32 copy against the exact projected systems (src/davidson.f90:719-732): the correction solves
    P (A32 - theta I) P t = -r, i.e. the fp64 system up to the rounding of the operator entries (6e-8 relative)."""
    n, m, L = 300, 6, 3
    A = O.generate_diagonal_dominant(n, 1e-2, seed=8)
    V = O.generate_preconditioner(np.diag(A).copy(), m)
    W = A @ V
    theta, Y = O.lapack_generalized_eigensolver(V.T @ W)
    X = V @ Y
    R = W @ Y - X * theta[None, :]
    out = {}
    for bits in (64, 32):
        with fd.CEngine(n=n, max_cols=2 * m) as e:
            e.set_storage(1)
            e.set_dense_host(OP_A, A)
            e.set_inner_precision(bits)
            e.panel_put(PANEL_V, 0, V)
            e.panel_put(PANEL_W, 0, W)
            e.ritz_residual_correction(m, L, Y, theta, METHOD_GJD)
            assert e.gjd_correction(m, theta, 300, 1e-10) > 0
            out[bits] = e.panel_get(PANEL_V, m, m)
    A32 = A.astype(np.float32).astype(np.float64)
    for k in range(m):
        x = X[:, k]
        P = np.eye(n) - np.outer(x, x)
        rk = max(1.0, np.linalg.norm(R[:, k]))
        M32 = P @ (A32 - theta[k] * np.eye(n)) @ P
        M64 = P @ (A - theta[k] * np.eye(n)) @ P
        assert np.linalg.norm(M32 @ out[32][:, k] + R[:, k]) < 1e-8 * rk           # solved what it was given ...
        assert np.linalg.norm(M64 @ out[32][:, k] + R[:, k]) < 1e-5 * rk           # ... which is the fp64 system to fp32 rounding
        assert np.linalg.norm(M64 @ out[64][:, k] + R[:, k]) < 1e-8 * rk
        assert np.linalg.norm(P @ (out[32][:, k] - out[64][:, k])) < 1e-5 * max(1e-30, np.linalg.norm(P @ out[64][:, k]))
    assert not np.array_equal(out[32], out[64])                                     # the fp32 path really ran


@pytest.mark.parametrize("name", GJD_CASES)
def test_gjd_golden_cases_with_fp32_inner_sweeps(golden, name):
    manifest, arrays = golden
    case = manifest["dense"][name]
    A, B = case_matrices(case, arrays)
    with fd.DavidsonEngine(case["n"], case["lowest"], case["max_dim"], gev=B is not None, storage="symmetric") as eng:
        eng.set_dense(1, A)
        if B is not None:
            eng.set_dense(2, B)
        eng.set_inner_precision(32)
        lam, vec, iters = eng.solve("GJD", case["max_it"], case["tol"])
    assert np.abs(lam - arrays[f"{name}__evals"]).max() < EV_TOL
    assert (residuals(A, B, lam, vec) < case["tol"]).all()
    assert iters == case["iters"]


def test_fp32_inner_sweeps_at_a_size_with_super_row_schedules(monkeypatch):
    """N=25000 generalized GJD (A and B stored as symmetric tiles), the environment knob of the drop-in front ends:
    same eigenvalues and the same outer iteration count as fp64 inner sweeps, fewer bytes per inner sweep."""
    n, L, sp = 25000, 4, 1e-3
    res = {}
    for bits in ("64", "32"):
        monkeypatch.setenv("DAVIDSON_INNER_PRECISION", bits)
        with fd.DavidsonEngine(n, L, gev=True, storage="symmetric") as eng:
            eng.generate_diagonal_dominant(1, sp, seed=1)
            eng.generate_diagonal_dominant(2, sp, 1.0, seed=2)
            eng.c.reset_stats()
            lam, _, iters = eng.solve("GJD", 100, 1e-8, want_vectors=False)
            st = eng.c.stats()
            res[bits] = (lam, iters, st.apply_bytes / max(st.applies, 1))
    assert res["32"][1] == res["64"][1]
    assert np.abs(res["32"][0] - res["64"][0]).max() < 1e-9
    assert res["32"][2] < 0.8 * res["64"][2]            # the inner sweeps (most of them) moved half the bytes


@pytest.mark.parametrize("n,L,sp,md,gev", [(200, 3, 5e-2, None, False), (300, 4, 1e-1, None, False),
                                           (200, 3, 5e-2, 10, True), (400, 5, 2e-2, None, True), (150, 2, 3e-1, 8, False)])
def test_fp32_inner_sweeps_on_harder_matrices_keep_the_oracle_iteration_counts(n, L, sp, md, gev, monkeypatch):
    """strong off-diagonals, generalized problems and restarts, through the drop-in call with both knobs from the
    environment (symmetric storage + fp32 inner sweeps): same outer iteration counts as the reference's dense DSYSV solves"""
    monkeypatch.setenv("DAVIDSON_STORAGE", "symmetric")
    monkeypatch.setenv("DAVIDSON_INNER_PRECISION", "32")
    A = O.generate_diagonal_dominant(n, sp, seed=5)
    B = O.generate_diagonal_dominant(n, sp * 0.1, 1.0, seed=6) if gev else None
    lam_o, _, it_o = O.generalized_eigensolver_dense(A, L, "GJD", 40, 1e-8, md, B)
    lam, vec, it = fd.generalized_eigensolver(A, L, "GJD", 40, 1e-8, md, B)
    assert it == it_o
    assert np.abs(lam - lam_o).max() < EV_TOL
    assert (residuals(A, B, lam, vec) < 1e-8).all()
