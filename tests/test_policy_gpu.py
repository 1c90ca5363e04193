"""Opt-in correction policies (not in the reference, so never the default): "unconverged" - corrections only
for the wanted Ritz pairs that have not converged - and "locking" - converged wanted pairs are locked, the search space is
kept orthogonal to them (the deflation the reference's header cites, src/davidson.f90:7-8).  Same eigenpairs as the
reference path (golden eigenvalues, residuals below the tolerance); iteration counts equal to the CPU statements of the
policies in oracle/davidson_oracle.py."""
import ctypes as C
import threading

import numpy as np
import pytest

import fortran_davidson_amd as fd
from fortran_davidson_amd.solver import generalized_eigensolver_free
from oracle import davidson_oracle as O
from conftest import case_matrices

pytestmark = pytest.mark.gpu
EV_TOL = 1e-8

CASES = ["matrix_txt_dpr", "matrix_txt_gjd", "c1_n50_std_dpr", "c1_n50_std_gjd", "c1_n50_gev_dpr", "c1_n50_gev_gjd",
         "n400_std_dpr", "n400_std_gjd", "n400_gev_gjd", "n1000_restart_dpr", "n1000_gev_restart_dpr", "n2000_std_dpr",
         "n3000_hard_dpr", "n4000_gev_dpr"]


def residuals(A, B, lam, X):
    BX = X if B is None else B @ X
    return np.linalg.norm(A @ X - BX * lam[None, :], axis=0)


@pytest.mark.parametrize("name", CASES)
def test_unconverged_policy_same_eigenpairs_and_oracle_iteration_count(golden, name):
    manifest, arrays = golden
    case = manifest["dense"][name]
    A, B = case_matrices(case, arrays)
    L, tol = case["lowest"], case["tol"]
    with fd.DavidsonEngine(case["n"], L, case["max_dim"], gev=B is not None) as eng:
        eng.set_correction_policy("unconverged")
        eng.set_dense(1, A)
        if B is not None:
            eng.set_dense(2, B)
        lam, vec, iters = eng.solve(case["method"], case["max_it"], tol)
        eng.set_correction_policy("all")                       # and back: the reference's path, golden iterations
        lam_all, _, iters_all = eng.solve(case["method"], case["max_it"], tol)
    assert np.abs(lam - arrays[f"{name}__evals"]).max() < EV_TOL
    assert (residuals(A, B, lam, vec) < tol).all()
    tr = O.Trace()
    lam_o, _, it_o = O.generalized_eigensolver_dense_unconverged(A, L, case["method"], case["max_it"], tol,
                                                                 case["max_dim"], B, trace=tr)
    assert iters == it_o, (iters, it_o, tr.widths)
    assert np.abs(lam - lam_o).max() < EV_TOL
    assert iters_all == case["iters"] and np.abs(lam_all - arrays[f"{name}__evals"]).max() < EV_TOL


LOCKING_CASES = [c for c in CASES if "gev" not in c]


@pytest.mark.parametrize("name", LOCKING_CASES)
def test_locking_policy_same_eigenpairs_and_oracle_iteration_count(golden, name):
    """"locking" on the standard golden cases: eigenvalues of the reference, residuals below the tolerance, orthonormal vectors, the
    iteration count of the oracle's statement of the policy; and back to the reference's policy on the same engine."""
    manifest, arrays = golden
    case = manifest["dense"][name]
    A, _ = case_matrices(case, arrays)
    L, tol = case["lowest"], case["tol"]
    with fd.DavidsonEngine(case["n"], L, case["max_dim"]) as eng:
        eng.set_correction_policy("locking")
        eng.set_dense(1, A)
        lam, vec, iters = eng.solve(case["method"], case["max_it"], tol)
        eng.set_correction_policy("all")
        lam_all, _, iters_all = eng.solve(case["method"], case["max_it"], tol)
    assert np.abs(lam - arrays[f"{name}__evals"]).max() < EV_TOL
    assert (residuals(A, None, lam, vec) < tol).all()
    assert np.abs(vec.T @ vec - np.eye(L)).max() < 1e-7          # locked vectors are Ritz vectors of different bases: orthogonal to ~tol
    lam_o, _, it_o = O.generalized_eigensolver_dense_locking(A, L, case["method"], case["max_it"], tol, case["max_dim"])
    assert iters == it_o, (iters, it_o)
    assert np.abs(lam - lam_o).max() < EV_TOL
    assert iters_all == case["iters"] and np.abs(lam_all - arrays[f"{name}__evals"]).max() < EV_TOL


@pytest.mark.parametrize("nranks,storage", [(1, "symmetric"), (3, "full"), (3, "symmetric")])
def test_locking_policy_through_restarts_on_one_and_three_ranks(nranks, storage):
    """A problem that locks its pairs at three different iterations and collapses the active basis on the way (lowest = 8,
    max_dim_sub = 28, a clustered and a spread part of the spectrum), on one rank and on three (loopback transport, both
    storages): the oracle's iteration count, its eigenvalues."""
    n, L, sp, md = 1500, 8, 3e-2, 28
    A = O.generate_diagonal_dominant(n, sp, seed=4)
    d = np.arange(1, n + 1, dtype=float) + 2.0
    d[:8] = [1.0, 1.08, 1.16, 2.5, 4.0, 4.1, 7.0, 9.5]          # a cluster and well separated values: the pairs converge at different times
    A[np.arange(n), np.arange(n)] = d
    tr = O.Trace()
    lam_o, _, it_o = O.generalized_eigensolver_dense_locking(A, L, "DPR", 300, 1e-8, md, trace=tr)
    assert it_o < 300 and len(set(len(e) for e in tr.errors)) > 2          # pairs were locked at different iterations
    engs = [fd.DavidsonEngine(n, L, md, rank=r, nranks=nranks, storage=storage) for r in range(nranks)]
    if nranks > 1:
        handles = (C.c_void_p * nranks)(*[e.c.h for e in engs])
        assert fd.hip_lib().dav_local_group_join(handles, nranks) == 0
    out, err = [None] * nranks, [None] * nranks

    def work(r):
        try:
            engs[r].set_correction_policy("locking")
            engs[r].set_dense(1, A)
            out[r] = engs[r].solve("DPR", 300, 1e-8)
        except Exception as exc:      # noqa: BLE001
            err[r] = exc

    threads = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in threads]
    [t.join(timeout=300) for t in threads]
    for e in engs:
        e.close()
    assert all(x is None for x in err), err
    for lam, vec, iters in out:
        assert iters == it_o, (iters, it_o)
        assert np.abs(lam - lam_o).max() < EV_TOL
        assert (residuals(A, None, lam, vec) < 1e-8).all()
        assert (np.diff(lam) > 0).all()


GEV_LOCKING_CASES = [(300, 3, 1e-2, None, "DPR", "full", 1), (500, 5, 3e-2, 15, "DPR", "symmetric", 1), (200, 2, 1e-2, None, "GJD", "full", 1),
                     (400, 8, 5e-2, 24, "DPR", "symmetric", 1), (700, 6, 3e-2, 20, "DPR", "full", 3), (700, 6, 3e-2, 20, "DPR", "symmetric", 3),
                     # pairs that lock while the basis is wider than 64 columns (the contraction then goes through the scratch panel)
                     (400, 12, 5e-2, 72, "DPR", "full", 1), (600, 16, 3e-2, 120, "DPR", "symmetric", 1)]


@pytest.mark.parametrize("n,L,sp,md,method,storage,nranks", GEV_LOCKING_CASES)
def test_locking_policy_on_generalized_problems(n, L, sp, md, method, storage, nranks):
    """"locking" for A x = lambda B x (round 6; the deflation of the paper the reference's header cites, src/davidson.f90:7-8, covers the
    generalized problem): a converged pair's eigenvector leaves the basis, its GUARD vector B x takes its place - the pairs still wanted
    are B-orthogonal to the locked ones, so the search space stays orthogonal to span(B X_locked) - and the Rayleigh-Ritz pencil is that
    of the active basis.  Against the oracle's statement of the policy (generalized_eigensolver_dense_locking with a second matrix):
    its iteration count, its eigenvalues; residuals below the tolerance, B-orthonormal eigenvectors, scipy's eigenvalues; one rank and
    three (loopback transport), both storages, DPR and GJD."""
    import scipy.linalg
    A = O.generate_diagonal_dominant(n, sp, seed=3)
    B = O.generate_diagonal_dominant(n, sp, 1.0, seed=4)
    if L >= 5:
        d = np.arange(1, n + 1, dtype=float) + 2.0
        d[:L] = np.cumsum([1.0] + [0.08 if i % 2 else 1.7 for i in range(L - 1)])        # pairs that converge at different iterations
        A[np.arange(n), np.arange(n)] = d
    tol = 1e-8
    lam_o, _, it_o = O.generalized_eigensolver_dense_locking(A, L, method, 200, tol, md, second_matrix=B)
    assert it_o < 200
    engs = [fd.DavidsonEngine(n, L, md, gev=True, rank=r, nranks=nranks, storage=storage) for r in range(nranks)]
    if nranks > 1:
        handles = (C.c_void_p * nranks)(*[e.c.h for e in engs])
        assert fd.hip_lib().dav_local_group_join(handles, nranks) == 0
    out, err = [None] * nranks, [None] * nranks

    def work(r):
        try:
            engs[r].set_correction_policy("locking")
            engs[r].set_dense(1, A)
            engs[r].set_dense(2, B)
            out[r] = engs[r].solve(method, 200, tol)
        except Exception as exc:      # noqa: BLE001
            err[r] = exc

    threads = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in threads]
    [t.join(timeout=300) for t in threads]
    for e in engs:
        e.close()
    assert all(x is None for x in err), err
    ref = scipy.linalg.eigh(A, B, eigvals_only=True, subset_by_index=[0, L - 1])
    for lam, vec, iters in out:
        assert iters == it_o, (iters, it_o)
        assert np.abs(lam - lam_o).max() < EV_TOL and np.abs(lam - ref).max() < EV_TOL
        assert (residuals(A, B, lam, vec) < tol).all()
        assert np.abs(vec.T @ B @ vec - np.eye(L)).max() < 1e-7
        assert (np.diff(lam) > 0).all()


def test_policy_through_the_environment_reaches_the_dense_and_free_front_ends(monkeypatch):
    n, L = 600, 4
    A = O.generate_diagonal_dominant(n, 2e-2, seed=8)
    B = O.generate_diagonal_dominant(n, 2e-2, 1.0, seed=9)
    lam_all, _, it_all = fd.generalized_eigensolver(A, L, "DPR", 200, 1e-8, 24)
    monkeypatch.setenv("DAVIDSON_CORRECTION_POLICY", "unconverged")
    lam, vec, it = fd.generalized_eigensolver(A, L, "DPR", 200, 1e-8, 24)
    lam_o, _, it_o = O.generalized_eigensolver_dense_unconverged(A, L, "DPR", 200, 1e-8, 24)
    assert it == it_o and np.abs(lam - lam_o).max() < EV_TOL and np.abs(lam - lam_all).max() < EV_TOL
    assert (residuals(A, None, lam, vec) < 1e-8).all()
    lam_f, vec_f, it_f = generalized_eigensolver_free(lambda x: A @ x, n, L, "DPR", 200, 1e-8, 24, lambda x: B @ x)
    import scipy.linalg
    ref = scipy.linalg.eigh(A, B, eigvals_only=True, subset_by_index=[0, L - 1])
    assert np.abs(lam_f - ref).max() < EV_TOL
    assert (residuals(A, B, lam_f, vec_f) < 1e-8).all()


def test_policy_with_symmetric_storage_and_restart():
    n, L = 3000, 8
    tr = O.Trace()
    A = O.generate_diagonal_dominant(n, 2e-2, seed=1)
    lam_o, _, it_o = O.generalized_eigensolver_dense_unconverged(A, L, "DPR", 200, 1e-8, 32, trace=tr)
    assert min(tr.widths[1:]) == 2 * L                         # the case does restart
    with fd.DavidsonEngine(n, L, 32, storage="symmetric") as eng:
        eng.set_correction_policy("unconverged")
        eng.generate_diagonal_dominant(1, 2e-2, seed=1)
        lam, vec, it = eng.solve("DPR", 200, 1e-8)
    assert it == it_o and np.abs(lam - lam_o).max() < EV_TOL
    assert (residuals(A, None, lam, vec) < 1e-8).all()


@pytest.mark.parametrize("nranks", [2, 3])
def test_policy_on_row_slab_ranks(golden, nranks):
    manifest, arrays = golden
    for name in ("n1000_gev_restart_dpr", "n400_gev_gjd"):
        case = manifest["dense"][name]
        A, B = case_matrices(case, arrays)
        _, _, it_o = O.generalized_eigensolver_dense_unconverged(A, case["lowest"], case["method"], case["max_it"],
                                                                 case["tol"], case["max_dim"], B)
        engs = [fd.DavidsonEngine(case["n"], case["lowest"], case["max_dim"], gev=True, rank=r, nranks=nranks)
                for r in range(nranks)]
        handles = (C.c_void_p * nranks)(*[e.c.h for e in engs])
        assert fd.hip_lib().dav_local_group_join(handles, nranks) == 0
        out = [None] * nranks

        def work(r):
            engs[r].set_correction_policy("unconverged")
            engs[r].generate_diagonal_dominant(1, case["sparsity"], seed=case["seed_a"])
            engs[r].generate_diagonal_dominant(2, case["sparsity"], 1.0, seed=case["seed_b"])
            out[r] = engs[r].solve(case["method"], case["max_it"], case["tol"])

        threads = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
        [t.start() for t in threads]
        [t.join(timeout=300) for t in threads]
        assert all(o is not None for o in out), "a rank did not finish"
        for lam, vec, iters in out:
            assert np.abs(lam - arrays[f"{name}__evals"]).max() < EV_TOL
            assert iters == it_o
            assert (residuals(A, B, lam, vec) < case["tol"]).all()
        for e in engs:
            e.close()


def test_panel_select_and_partial_ritz_phase_against_numpy():
    """The two C-ABI pieces the policy adds, on their own."""
    from fortran_davidson_amd.engine_c import OP_A, PANEL_V, PANEL_W, PANEL_X
    n, m, L = 700, 12, 5
    rng = np.random.default_rng(2)
    A = O.generate_diagonal_dominant(n, 1e-2, seed=4)
    V, _ = np.linalg.qr(rng.standard_normal((n, m)))
    with fd.CEngine(n=n, max_cols=64) as e:
        e.set_dense_host(OP_A, A)
        e.panel_put(PANEL_V, 0, V)
        e.apply(OP_A, PANEL_V, 0, m, PANEL_W, 0)
        e.set_width(m)
        H = V.T @ A @ V
        theta, Y = np.linalg.eigh(H)
        lib = fd.hip_lib()
        res = np.zeros(L)
        Yl = np.asfortranarray(Y[:, :L])
        rc = lib.dav_ritz_residual_correction_n(e.h, C.c_int(m), C.c_int(L), C.c_int(L), Yl.ctypes.data_as(C.POINTER(C.c_double)),
                                                C.c_int64(m), theta[:L].ctypes.data_as(C.POINTER(C.c_double)), C.c_int(0),
                                                res.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc == 0, lib.dav_last_error()
        X = V @ Y[:, :L]
        R = A @ X - X * theta[None, :L]
        assert np.allclose(res, np.linalg.norm(R, axis=0), rtol=1e-10)
        assert np.allclose(e.panel_get(PANEL_X, 0, L), X, atol=1e-12)
        T = R / (theta[None, :L] - np.diag(A)[:, None])
        assert np.allclose(e.panel_get(PANEL_V, m, L), T, rtol=1e-9, atol=1e-12)
        sel = np.array([1, 3, 4], dtype=np.int32)
        assert lib.dav_panel_select(e.h, C.c_int(PANEL_V), C.c_int(m), C.c_int(3), sel.ctypes.data_as(C.POINTER(C.c_int))) == 0
        assert np.allclose(e.panel_get(PANEL_V, m, 3), T[:, sel], rtol=1e-9, atol=1e-12)
        bad = np.array([2, 1], dtype=np.int32)
        assert lib.dav_panel_select(e.h, C.c_int(PANEL_V), C.c_int(m), C.c_int(2), bad.ctypes.data_as(C.POINTER(C.c_int))) != 0
        assert b"ascending" in lib.dav_last_error()
