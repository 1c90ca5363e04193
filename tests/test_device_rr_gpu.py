"""Device-side Rayleigh-Ritz: the one-workgroup Jacobi eigensolver against LAPACK (scipy: DSYEVD / DSYGVD
- what the host path calls, src/lapack_wrapper.f90:14-91) on the projected matrices of real bases, and whole solves with
the eigenpairs kept on the device against the reference's golden values: same eigenvalues, same iteration counts."""
import numpy as np
import pytest
import scipy.linalg

import fortran_davidson_amd as fd
from fortran_davidson_amd.engine_c import OP_A, OP_B, PANEL_V, PANEL_W, PANEL_BV, PANEL_X, PANEL_R, METHOD_DPR
from oracle import davidson_oracle as O
from conftest import case_matrices
from test_solver_gpu import DPR_CASES, GJD_CASES, residuals, EV_TOL

pytestmark = pytest.mark.gpu
METHOD_NONE = 2


@pytest.mark.parametrize("gev", [False, True])
@pytest.mark.parametrize("m", [1, 2, 5, 16, 33, 64, 96, 97, 128])
def test_small_eigensolver_against_lapack(m, gev):
    """eigenvalues to 1e-12 (relative to each eigenvalue's own size: the Jacobi criterion is entrywise-relative),
    eigenvectors through the residual H y - theta S y and the normalisation Y^T S Y = I."""
    n = 700
    rng = np.random.default_rng(100 * m + gev)
    # graded operator: diagonal 1..n plus noise, so that the projected matrix spans orders of magnitude like a Davidson basis
    A = np.diag(np.arange(1.0, n + 1)) * (1.0 + 300.0 * (np.arange(n) > 40)) + 1e-2 * rng.standard_normal((n, n))
    A = 0.5 * (A + A.T)
    B = np.eye(n) + 1e-2 * rng.standard_normal((n, n)); B = 0.5 * (B + B.T)
    V = np.linalg.qr(rng.standard_normal((n, m)))[0]
    with fd.CEngine(n=n, max_cols=128, gev=gev) as e:
        e.set_dense_host(OP_A, A)
        if gev:
            e.set_dense_host(OP_B, B)
        e.rr_enable(True)
        e.panel_put(PANEL_V, 0, V)
        e.apply(OP_A, PANEL_V, 0, m, PANEL_W, 0)
        if gev:
            e.apply(OP_B, PANEL_V, 0, m, PANEL_BV, 0)
        H = np.zeros((m, m), order="F"); S = np.zeros((m, m), order="F")
        e.project(0, m, H, S if gev else None)             # host copies AND the device-resident ones
        L = min(3, m)
        theta, res, sweeps = e.rr_ritz(m, m, L, METHOD_NONE)
        th2, Y = e.rr_get(m, m)
        X = e.panel_get(PANEL_X, 0, L)
    assert np.array_equal(theta, th2) and 0 < sweeps <= 12 or m == 1
    ref = scipy.linalg.eigh(H, S if gev else None, eigvals_only=True)
    assert np.abs(theta - ref).max() <= 1e-12 * np.abs(ref).max()
    assert (np.abs(theta - ref) <= 1e-11 * np.abs(ref) + 1e-13).all()       # the small ones keep their digits
    Sm = S if gev else np.eye(m)
    assert np.abs(H @ Y - Sm @ Y * theta[None, :]).max() <= 1e-11 * np.abs(H).max()
    assert np.abs(Y.T @ Sm @ Y - np.eye(m)).max() <= 1e-12
    # the fused Ritz phase used the device-resident eigenpairs: X = V Y(:, :L), residual norms of the first L pairs
    assert np.abs(X - V @ Y[:, :L]).max() <= 1e-12
    BX = (B if gev else np.eye(n)) @ X
    assert np.allclose(res, np.linalg.norm(A @ X - BX * theta[None, :L], axis=0), rtol=1e-9, atol=1e-9)


def test_projected_overlap_that_is_not_positive_definite_is_an_error():
    n, m = 300, 8
    rng = np.random.default_rng(0)
    A = np.diag(np.arange(1.0, n + 1))
    B = -np.eye(n)
    V = np.linalg.qr(rng.standard_normal((n, m)))[0]
    with fd.CEngine(n=n, max_cols=16, gev=True) as e:
        e.set_dense_host(OP_A, A); e.set_dense_host(OP_B, B)
        e.rr_enable(True)
        e.panel_put(PANEL_V, 0, V)
        e.apply(OP_A, PANEL_V, 0, m, PANEL_W, 0); e.apply(OP_B, PANEL_V, 0, m, PANEL_BV, 0)
        e.project_dev(0, m)
        with pytest.raises(fd.DavidsonHipError, match="positive definite"):
            e.rr_ritz(m, m, 2, METHOD_NONE)


@pytest.mark.parametrize("name", DPR_CASES + GJD_CASES)
def test_solves_with_device_rr_match_reference_golden(golden, name, monkeypatch):
    """the drop-in entry point with DAVIDSON_DEVICE_RR=1: golden eigenvalues, residuals, iteration counts"""
    monkeypatch.setenv("DAVIDSON_DEVICE_RR", "1")
    manifest, arrays = golden
    case = manifest["dense"][name]
    A, B = case_matrices(case, arrays)
    lam, vec, iters = fd.generalized_eigensolver(A, case["lowest"], case["method"], case["max_it"], case["tol"],
                                                 case["max_dim"], B)
    assert np.abs(lam - arrays[f"{name}__evals"]).max() < EV_TOL
    assert (residuals(A, B, lam, vec) < case["tol"]).all()
    assert iters == case["iters"]
    Bm = np.eye(A.shape[0]) if B is None else B
    assert np.allclose(vec.T @ Bm @ vec, np.eye(case["lowest"]), atol=1e-10)


@pytest.mark.parametrize("storage,policy,nranks", [("symmetric", "all", 1), ("full", "unconverged", 1), ("symmetric", "all", 3)])
def test_device_rr_with_restarts_policies_and_ranks(golden, storage, policy, nranks):
    import ctypes as C
    import threading
    manifest, arrays = golden
    for name in ("n1000_restart_dpr", "n1000_gev_restart_dpr"):
        case = manifest["dense"][name]
        engs = [fd.DavidsonEngine(case["n"], case["lowest"], case["max_dim"], gev=case["gev"], rank=r, nranks=nranks, storage=storage)
                for r in range(nranks)]
        if nranks > 1:
            handles = (C.c_void_p * nranks)(*[e.c.h for e in engs])
            assert fd.hip_lib().dav_local_group_join(handles, nranks) == 0
        out = [None] * nranks

        def work(r):
            eng = engs[r]
            eng.generate_diagonal_dominant(1, case["sparsity"], seed=case["seed_a"])
            if case["gev"]:
                eng.generate_diagonal_dominant(2, case["sparsity"], 1.0, seed=case["seed_b"])
            eng.set_correction_policy(policy)
            host = eng.solve(case["method"], case["max_it"], case["tol"])
            eng.set_device_rr(True)
            dev = eng.solve(case["method"], case["max_it"], case["tol"])
            out[r] = (host, dev)

        threads = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
        [t.start() for t in threads]
        [t.join(timeout=300) for t in threads]
        assert all(o is not None for o in out)
        A, B = case_matrices(case, arrays)
        for host, dev in out:
            assert dev[2] == host[2]                                        # same iteration count as the host path
            assert np.abs(dev[0] - host[0]).max() < 1e-10
            assert np.abs(dev[0] - arrays[f"{name}__evals"]).max() < EV_TOL
            assert (residuals(A, B, dev[0], dev[1]) < case["tol"]).all()
        for e in engs:
            e.close()


@pytest.mark.parametrize("n,L,sp,md", [(17, 1, 1e-2, None), (33, 2, 1e-2, None), (64, 3, 1e-2, 4), (130, 5, 5e-2, 12),
                                       (257, 1, 1e-1, 3), (500, 16, 1e-2, None)])
def test_edge_shapes_with_device_rr_against_oracle(n, L, sp, md, monkeypatch):
    """tiny and ragged orders, lowest = 1 (projected problems of order 2), restarts every iteration, a basis of 128 columns:
    the device eigensolver drives the outer loop exactly like host LAPACK (and the oracle)"""
    monkeypatch.setenv("DAVIDSON_DEVICE_RR", "1")
    A = O.generate_diagonal_dominant(n, sp, seed=31)
    tr = O.Trace()
    lam_o, vec_o, it_o = O.generalized_eigensolver_dense(A, L, "DPR", 60, 1e-8, md, trace=tr)
    lam, vec, it = fd.generalized_eigensolver(A, L, "DPR", 60, 1e-8, md)
    assert it == it_o, (it, it_o, tr.widths)
    assert np.abs(lam - lam_o).max() < EV_TOL
    if tr.converged:
        assert (residuals(A, None, lam, vec) < 1e-8).all()
