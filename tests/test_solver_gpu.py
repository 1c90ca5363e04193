"""GPU parity proper: the Fortran API (`generalized_eigensolver`) running on the HIP engine against
the golden vectors produced by the reference and against the numpy oracle on the same inputs.

Bar (BASELINE.json): eigenvalues within 1e-8 of the reference path, Ritz residuals below the
tolerance on both; we additionally require the iteration count to match."""
import numpy as np
import pytest

import fortran_davidson_amd as fd
from fortran_davidson_amd.solver import generalized_eigensolver_free
from oracle import davidson_oracle as O
from conftest import case_matrices

pytestmark = pytest.mark.gpu
EV_TOL = 1e-8

DPR_CASES = ["matrix_txt_dpr", "c1_n50_std_dpr", "c1_n50_gev_dpr", "n100_main_gev_dpr", "n400_std_dpr",
             "n1000_restart_dpr", "n1000_gev_restart_dpr", "n2000_std_dpr", "n3000_hard_dpr", "n4000_gev_dpr"]


def residuals(A, B, lam, X):
    BX = X if B is None else B @ X
    return np.linalg.norm(A @ X - BX * lam[None, :], axis=0)


@pytest.mark.parametrize("name", DPR_CASES)
def test_dense_dpr_matches_reference_golden(golden, name):
    manifest, arrays = golden
    case = manifest["dense"][name]
    A, B = case_matrices(case, arrays)
    lam, vec, iters = fd.generalized_eigensolver(A, case["lowest"], case["method"], case["max_it"], case["tol"],
                                                 case["max_dim"], B)
    assert np.abs(lam - arrays[f"{name}__evals"]).max() < EV_TOL
    assert np.allclose(lam, case["eigh"])                       # the reference's own checker criterion
    assert (residuals(A, B, lam, vec) < case["tol"]).all()
    assert iters == case["iters"]
    if B is None:
        assert np.allclose(vec.T @ vec, np.eye(case["lowest"]), atol=1e-10)
    else:
        assert np.allclose(vec.T @ B @ vec, np.eye(case["lowest"]), atol=1e-10)   # DSYGV itype=1 normalisation


@pytest.mark.parametrize("gev", [False, True])
def test_gjd_outer_iterations_equal_the_oracles_exact_solves_on_a_grid(gev):
    """The engine solves the GJD correction equations inexactly (block MINRES, wanted pairs to 0.01 tol / |r|, the others to 1e-2:
    fortran/davidson.f90 gjd_tol_wanted / gjd_tol_unwanted); the reference solves them exactly (DSYSV, src/davidson.f90:700-734).
    Same outer iteration count, eigenvalues and residuals as the oracle's restatement of the reference on a grid of problems of
    our generator (tests/gjd_policy_sweep.py runs the larger grid and the alternative settings)."""
    for n, lowest, sp, seed in [(150, 2, 1e-3, 1), (150, 4, 1e-2, 2), (150, 8, 1e-2, 1), (300, 2, 1e-2, 2), (300, 4, 1e-3, 1),
                                (300, 8, 1e-2, 2), (500, 4, 1e-2, 1), (500, 8, 1e-3, 2)]:
        A = O.generate_diagonal_dominant(n, sp, seed=seed)
        B = O.generate_diagonal_dominant(n, sp, 1.0, seed=seed + 100) if gev else None
        lam_o, _, it_o = O.generalized_eigensolver_dense(A, lowest, "GJD", 50, 1e-8, None, B)
        lam, vec, it = fd.generalized_eigensolver(A, lowest, "GJD", 50, 1e-8, None, B)
        assert it == it_o, (n, lowest, sp, seed, it, it_o)
        assert np.abs(lam - lam_o).max() < 1e-8
        assert (residuals(A, B, lam, vec) < 1e-8).all()


GJD_CASES = ["matrix_txt_gjd", "c1_n50_std_gjd", "c1_n50_gev_gjd", "n100_main_gev_gjd", "n400_std_gjd", "n400_gev_gjd"]


@pytest.mark.parametrize("name", GJD_CASES)
def test_dense_gjd_matches_reference_golden(golden, name):
    """GJD: the reference solves the projected systems densely (DSYSV); the engine solves the same
    systems with block MINRES on the device.  Same eigenvalues, residuals below tolerance, same
    number of outer iterations."""
    manifest, arrays = golden
    case = manifest["dense"][name]
    A, B = case_matrices(case, arrays)
    lam, vec, iters = fd.generalized_eigensolver(A, case["lowest"], "GJD", case["max_it"], case["tol"],
                                                 case["max_dim"], B)
    assert np.abs(lam - arrays[f"{name}__evals"]).max() < EV_TOL
    assert (residuals(A, B, lam, vec) < case["tol"]).all()
    assert iters == case["iters"]


def test_gjd_correction_solves_the_projected_systems():
    """K7 against the oracle's dense solve (davidson.f90:700-734) on one block."""
    from fortran_davidson_amd.engine_c import OP_A, PANEL_V, PANEL_W, PANEL_R, METHOD_GJD
    n, m, L = 300, 6, 3
    A = O.generate_diagonal_dominant(n, 1e-2, seed=8)
    V = O.generate_preconditioner(np.diag(A).copy(), m)
    W = A @ V
    theta, Y = O.lapack_generalized_eigensolver(V.T @ W)
    X = V @ Y
    R = W @ Y - X * theta[None, :]
    T_ref = O.compute_GJD_generalized_dense(A, theta, X, R)
    with fd.CEngine(n=n, max_cols=2 * m) as e:
        e.set_dense_host(OP_A, A)
        e.panel_put(PANEL_V, 0, V)
        e.panel_put(PANEL_W, 0, W)
        e.ritz_residual_correction(m, L, Y, theta, METHOD_GJD)
        its = e.gjd_correction(m, theta, 300, 1e-10)
        T = e.panel_get(PANEL_V, m, m)
    assert its > 0
    for k in range(m):
        x = X[:, k]
        P = np.eye(n) - np.outer(x, x)
        Mk = P @ (A - theta[k] * np.eye(n)) @ P
        # both solve the singular consistent system; compare after removing the null-space component
        assert np.linalg.norm(Mk @ T[:, k] + R[:, k]) < 1e-8 * max(1.0, np.linalg.norm(R[:, k]))
        assert np.linalg.norm(P @ T[:, k] - P @ T_ref[:, k]) < 1e-6 * max(1e-30, np.linalg.norm(P @ T_ref[:, k]))


def test_gjd_correction_per_column_tolerances_and_followers():
    """dav_gjd_correction_n: every column stops at its own relative tolerance; a column with a NEGATIVE tolerance follows - it stops at
    |tol| or as soon as every column with a positive tolerance has stopped, whichever comes first (include/davidson_hip.h).  The
    residual of each projected system (src/davidson.f90:719-732) shows where its column stopped."""
    from fortran_davidson_amd.engine_c import OP_A, PANEL_V, PANEL_W, PANEL_R, METHOD_GJD
    n, m = 600, 8
    A = O.generate_diagonal_dominant(n, 2e-2, seed=9)
    V = O.generate_preconditioner(np.diag(A).copy(), m)
    W = A @ V
    theta, Y = O.lapack_generalized_eigensolver(V.T @ W)
    X = V @ Y
    R = W @ Y - X * theta[None, :]

    def run(tols):
        with fd.CEngine(n=n, max_cols=2 * m) as e:
            e.set_dense_host(OP_A, A)
            e.panel_put(PANEL_V, 0, V)
            e.panel_put(PANEL_W, 0, W)
            e.ritz_residual_correction(m, m, Y, theta, METHOD_GJD)
            its = e.gjd_correction_n(m, m, theta, tols, 300, 1e-10)
            T = e.panel_get(PANEL_V, m, m)
        res = []
        for k in range(m):
            P = np.eye(n) - np.outer(X[:, k], X[:, k])
            res.append(np.linalg.norm(P @ (A - theta[k] * np.eye(n)) @ P @ T[:, k] + R[:, k]) / np.linalg.norm(R[:, k]))
        return its, np.array(res)

    its_all, res_all = run(np.full(m, 1e-10))
    assert (res_all < 1e-7).all()
    # the second half only to 1e-2: those columns stop early, the others are solved as before
    its_mixed, res_mixed = run(np.r_[np.full(4, 1e-10), np.full(4, 1e-2)])
    assert (res_mixed[:4] < 1e-7).all() and (res_mixed[4:] < 0.2).all() and res_mixed[4:].max() > 1e-6
    # followers of a leader that stops at once (tolerance 0.9): everything stops within a step or two, long before 1e-10
    its_follow, res_follow = run(np.r_[np.full(4, 0.9), np.full(4, -1e-10)])
    assert its_follow < its_all and its_follow <= 3 and res_follow[4:].min() > 1e-8
    # followers stop by themselves at |tol| when the leaders take longer
    its_f2, res_f2 = run(np.r_[np.full(4, 1e-10), np.full(4, -1e-2)])
    assert its_f2 == its_mixed and np.allclose(res_f2, res_mixed)


def test_device_resident_engine_and_generator(golden):
    manifest, arrays = golden
    case = manifest["dense"]["n2000_std_dpr"]
    with fd.DavidsonEngine(case["n"], case["lowest"]) as eng:
        eng.generate_diagonal_dominant(1, case["sparsity"], seed=case["seed_a"])
        for _ in range(2):                                       # operators stay resident across solves
            lam, vec, iters = eng.solve("DPR", 1000, 1e-8)
            assert np.abs(lam - arrays["n2000_std_dpr__evals"]).max() < EV_TOL
            assert iters == case["iters"]
        st = eng.c.stats()
        assert st.applies > 0 and st.apply_ms > 0
        lam2, none, _ = eng.solve("DPR", 1000, 1e-8, want_vectors=False)
        assert none is None and np.array_equal(lam2, lam)


@pytest.mark.parametrize("name", ["n2000_std_dpr", "n1000_gev_restart_dpr", "n3000_hard_dpr"])
def test_symmetric_tiled_storage_solves_match_golden(golden, name):
    """Same solves with only the lower block triangle resident (K1s sweep)."""
    manifest, arrays = golden
    case = manifest["dense"][name]
    with fd.DavidsonEngine(case["n"], case["lowest"], case["max_dim"], gev=case["gev"], storage="symmetric") as eng:
        eng.generate_diagonal_dominant(1, case["sparsity"], seed=case["seed_a"])
        if case["gev"]:
            eng.generate_diagonal_dominant(2, case["sparsity"], 1.0, seed=case["seed_b"])
        lam, vec, iters = eng.solve(case["method"], case["max_it"], case["tol"])
    assert np.abs(lam - arrays[f"{name}__evals"]).max() < EV_TOL
    assert iters == case["iters"]
    A, B = case_matrices(case, arrays)
    assert (residuals(A, B, lam, vec) < case["tol"]).all()


def test_symmetric_storage_upload_and_gjd(golden):
    manifest, arrays = golden
    case = manifest["dense"]["n400_std_gjd"]
    A, _ = case_matrices(case, arrays)
    with fd.DavidsonEngine(case["n"], case["lowest"], storage="symmetric") as eng:
        eng.set_dense(1, A)
        lam, vec, iters = eng.solve("GJD", case["max_it"], case["tol"])
    assert np.abs(lam - arrays["n400_std_gjd__evals"]).max() < EV_TOL
    assert iters == case["iters"]


def test_gev_engine_hashed_operator_equals_dense(golden):
    manifest, arrays = golden
    case = manifest["dense"]["n1000_gev_restart_dpr"]
    with fd.DavidsonEngine(case["n"], case["lowest"], gev=True) as eng:
        eng.set_hashed_operator(1, case["sparsity"], seed=case["seed_a"])
        eng.set_hashed_operator(2, case["sparsity"], 1.0, seed=case["seed_b"])
        lam, vec, iters = eng.solve("DPR", 1000, 1e-8)
    assert np.abs(lam - arrays["n1000_gev_restart_dpr__evals"]).max() < EV_TOL
    assert iters == case["iters"]


@pytest.mark.parametrize("name", ["free_n50", "free_n300"])
def test_matrix_free_callbacks_match_reference_golden(golden, name):
    manifest, arrays = golden
    case = manifest["free"][name]
    n = case["n"]
    mtx, stx = O.harness_matrices(n)
    lam, vec, iters = generalized_eigensolver_free(lambda x: mtx @ x, n, case["lowest"], "DPR", case["max_it"],
                                                   case["tol"], case["max_dim"], lambda x: stx @ x)
    assert np.abs(lam - arrays[f"{name}__evals"]).max() < 1e-8
    assert np.allclose(lam, case["eigh"])
    assert iters == case["iters"]
    assert (residuals(mtx, stx, lam, vec) < case["tol"]).all()


@pytest.mark.parametrize("name", ["free_n50", "free_n300"])
def test_matrix_free_device_harness_operator(golden, name):
    manifest, arrays = golden
    case = manifest["free"][name]
    with fd.DavidsonEngine(case["n"], case["lowest"], case["max_dim"], gev=True) as eng:
        eng.set_harness_operator(1)
        eng.set_harness_operator(2)
        lam, vec, iters = eng.solve("DPR", case["max_it"], case["tol"])
    assert np.abs(lam - arrays[f"{name}__evals"]).max() < 1e-8
    assert iters == case["iters"]             # operator A is matrix-free: the matrix-free driver's convergence test


def test_benchmark_free_shape_identity_b():
    """src/benchmark_free.f90:80-111: N=1000, lowest=3, max_dim 20, B = I."""
    n = 1000
    mtx, _ = O.harness_matrices(n)
    with fd.DavidsonEngine(n, 3, 20, gev=True) as eng:
        eng.set_harness_operator(1)
        eng.set_identity(2)
        lam, vec, iters = eng.solve("DPR", 1000, 1e-8)
    ref = np.linalg.eigvalsh(mtx)[:3]
    assert np.abs(lam - ref).max() < 1e-8
    assert (residuals(mtx, None, lam, vec) < 1e-8).all()


def test_non_convergence_reports_max_it_plus_one():
    A = O.generate_diagonal_dominant(300, 5e-2, seed=4)
    lam, vec, iters = fd.generalized_eigensolver(A, 4, "DPR", 2, 1e-12)
    assert iters == 3                                   # src/davidson.f90:232-235
    lam_o, _, it_o = O.generalized_eigensolver_dense(A, 4, "DPR", 2, 1e-12)
    assert it_o == 3 and np.abs(lam - lam_o).max() < 1e-8


def test_against_oracle_on_fresh_inputs():
    for n, L, sp, seed in [(257, 2, 1e-2, 21), (900, 5, 1e-2, 22), (1500, 6, 3e-2, 23)]:
        A = O.generate_diagonal_dominant(n, sp, seed=seed)
        tr = O.Trace()
        lam_o, vec_o, it_o = O.generalized_eigensolver_dense(A, L, "DPR", 300, 1e-8, trace=tr)
        lam, vec, it = fd.generalized_eigensolver(A, L, "DPR", 300, 1e-8)
        assert it == it_o
        assert np.abs(lam - lam_o).max() < EV_TOL
        assert (residuals(A, None, lam, vec) < 1e-8).all()


@pytest.mark.parametrize("storage,direct", [("full", "0"), ("symmetric", "0"), ("symmetric", "1")])
def test_sharded_code_path_through_rccl_single_rank(golden, monkeypatch, storage, direct):
    """DAVIDSON_FORCE_RCCL=1: a 1-rank RCCL communicator, so the all-gather of the packed basis block,
    the all-reduces of Gram blocks / norms, the gathered panel download and - symmetric storage - the reduce-scatter
    of the partial products all run through RCCL.  direct = 1 (DAV_COLL_DIRECT, opt-in): the all-gather and the reduce-scatter as
    grouped send / receive exchanges with a fixed-order local sum - with one rank that is the plumbing only (no peer to talk to)."""
    monkeypatch.setenv("DAVIDSON_FORCE_RCCL", "1")
    monkeypatch.setenv("DAV_COLL_DIRECT", direct)
    manifest, arrays = golden
    for name in ("n2000_std_dpr", "n1000_gev_restart_dpr"):
        case = manifest["dense"][name]
        with fd.DavidsonEngine(case["n"], case["lowest"], case["max_dim"], gev=case["gev"], storage=storage) as eng:
            eng.comm_init(fd.CEngine.comm_unique_id())
            eng.c.set_timing(2)                                   # time the collectives too
            eng.generate_diagonal_dominant(1, case["sparsity"], seed=case["seed_a"])
            if case["gev"]:
                eng.generate_diagonal_dominant(2, case["sparsity"], 1.0, seed=case["seed_b"])
            lam, vec, iters = eng.solve("DPR", case["max_it"], case["tol"])
            assert eng.c.stats().comm_ms > 0
        assert np.abs(lam - arrays[f"{name}__evals"]).max() < EV_TOL
        assert iters == case["iters"]
        A, B = case_matrices(case, arrays)
        assert (residuals(A, B, lam, vec) < case["tol"]).all()


@pytest.mark.parametrize("n,L,sp,md", [(17, 1, 1e-2, None), (33, 2, 1e-2, None), (64, 3, 1e-2, 4), (130, 5, 5e-2, 12),
                                       (257, 1, 1e-1, 3), (500, 16, 1e-2, None)])
def test_edge_shapes_against_oracle(n, L, sp, md):
    """Tiny and ragged orders, lowest=1, max_dim_sub below the initial width (restart every iteration),
    width not a multiple of the MFMA tile: same eigenvalues, iteration counts and width policy."""
    A = O.generate_diagonal_dominant(n, sp, seed=31)
    tr = O.Trace()
    lam_o, vec_o, it_o = O.generalized_eigensolver_dense(A, L, "DPR", 60, 1e-8, md, trace=tr)
    lam, vec, it = fd.generalized_eigensolver(A, L, "DPR", 60, 1e-8, md)
    assert it == it_o, (it, it_o, tr.widths)
    assert np.abs(lam - lam_o).max() < EV_TOL
    if tr.converged:
        assert (residuals(A, None, lam, vec) < 1e-8).all()


def test_duplicate_diagonal_entries_use_stable_order():
    """Ties in the diagonal: the reference's key search is undefined; engine and
    oracle both take the stable order."""
    n, L = 200, 3
    A = O.generate_diagonal_dominant(n, 1e-3, seed=9)
    d = np.arange(1, n + 1, dtype=float)
    d[:8] = [2.0, 1.0, 2.0, 1.0, 3.0, 3.0, 1.0, 2.0]
    A[np.arange(n), np.arange(n)] = d
    lam_o, _, it_o = O.generalized_eigensolver_dense(A, L, "DPR", 100, 1e-8)
    lam, vec, it = fd.generalized_eigensolver(A, L, "DPR", 100, 1e-8)
    assert it == it_o and np.abs(lam - lam_o).max() < EV_TOL
    assert (residuals(A, None, lam, vec) < 1e-8).all()


def test_generalized_matrix_free_with_callbacks_and_general_b():
    """Host callbacks with a non-trivial B (dense numpy operators), checked against scipy."""
    import scipy.linalg
    n, L = 300, 4
    A = O.generate_diagonal_dominant(n, 5e-3, seed=41)
    B = O.generate_diagonal_dominant(n, 5e-3, 1.0, seed=42)
    lam, vec, it = generalized_eigensolver_free(lambda x: A @ x, n, L, "DPR", 200, 1e-8, 40, lambda x: B @ x)
    ref = scipy.linalg.eigh(A, B, eigvals_only=True, subset_by_index=[0, L - 1])
    assert np.abs(lam - ref).max() < 1e-8
    assert (residuals(A, B, lam, vec) < 1e-8).all()
    lam_o, _, it_o = O.generalized_eigensolver_free(lambda x: A @ x, n, L, 200, 1e-8, 40, lambda x: B @ x,
                                                    diag_matrix=np.diag(A).copy(), diag_second_matrix=np.diag(B).copy())
    assert it == it_o


@pytest.mark.parametrize("storage", ["full", "symmetric"])
@pytest.mark.parametrize("nranks", [2, 3])
def test_multi_rank_engine_on_one_gpu_through_loopback_transport(golden, nranks, storage):
    """The row-slab engine with nranks > 1, every rank a thread of this process on the same GPU,
    collectives through the loopback transport (device copies + barriers; RCCL semantics).  Checks the
    slab offsets, padded gathers, gathered diagonal / top-k selection and the gathered eigenvectors."""
    import ctypes as C
    import threading
    manifest, arrays = golden
    for name in ("n1000_restart_dpr", "n1000_gev_restart_dpr", "n400_gev_gjd"):
        case = manifest["dense"][name]
        engs = [fd.DavidsonEngine(case["n"], case["lowest"], case["max_dim"], gev=case["gev"], rank=r, nranks=nranks, storage=storage)
                for r in range(nranks)]
        handles = (C.c_void_p * nranks)(*[e.c.h for e in engs])
        assert fd.hip_lib().dav_local_group_join(handles, nranks) == 0
        out = [None] * nranks

        def work(r):
            eng = engs[r]
            eng.generate_diagonal_dominant(1, case["sparsity"], seed=case["seed_a"])
            if case["gev"]:
                eng.generate_diagonal_dominant(2, case["sparsity"], 1.0, seed=case["seed_b"])
            out[r] = eng.solve(case["method"], case["max_it"], case["tol"])

        threads = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
        [t.start() for t in threads]
        [t.join(timeout=300) for t in threads]
        assert all(o is not None for o in out), "a rank did not finish"
        A, B = case_matrices(case, arrays)
        for r in range(nranks):
            lam, vec, iters = out[r]
            assert np.abs(lam - arrays[f"{name}__evals"]).max() < EV_TOL
            assert iters == case["iters"]
            assert (residuals(A, B, lam, vec) < case["tol"]).all()
            assert np.array_equal(lam, out[0][0])               # every rank holds the same answer
        for e in engs:
            e.close()


@pytest.mark.parametrize("storage", ["full", "symmetric"])
def test_multi_rank_matrix_free_operator_through_loopback_transport(storage):
    """configs[4] shape in miniature: hashed matrix-free operator, B = I, over 4 ranks - row slabs (every rank generates
    its rows) or symmetric generation (every rank generates the lower-triangle tiles of its block rows once)."""
    import ctypes as C
    import threading
    n, L, sp, nranks = 3000, 4, 3e-3, 4
    engs = [fd.DavidsonEngine(n, L, gev=True, rank=r, nranks=nranks, storage=storage) for r in range(nranks)]
    handles = (C.c_void_p * nranks)(*[e.c.h for e in engs])
    assert fd.hip_lib().dav_local_group_join(handles, nranks) == 0
    out = [None] * nranks

    def work(r):
        engs[r].set_hashed_operator(1, sp, seed=3)
        engs[r].set_identity(2)
        out[r] = engs[r].solve("DPR", 200, 1e-8)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in threads]
    [t.join(timeout=300) for t in threads]
    assert all(o is not None for o in out)
    A = O.generate_diagonal_dominant(n, sp, seed=3)
    lam_o, _, it_o = O.generalized_eigensolver_dense(A, L, "DPR", 200, 1e-8)
    for lam, vec, iters in out:
        assert iters == it_o and np.abs(lam - lam_o).max() < EV_TOL
        assert (residuals(A, None, lam, vec) < 1e-8).all()
    for e in engs:
        e.close()


@pytest.mark.parametrize("n,L,sp,md,gev", [(200, 3, 5e-2, None, False), (300, 4, 1e-1, None, False),
                                           (200, 3, 5e-2, 10, True), (400, 5, 2e-2, None, True), (150, 2, 3e-1, 8, False)])
def test_gjd_on_harder_matrices_matches_oracle_iteration_counts(n, L, sp, md, gev):
    """Strong off-diagonals, generalized problems and restarts: the device MINRES corrections must drive
    the outer iteration exactly like the reference's dense DSYSV solves."""
    A = O.generate_diagonal_dominant(n, sp, seed=5)
    B = O.generate_diagonal_dominant(n, sp * 0.1, 1.0, seed=6) if gev else None
    lam_o, _, it_o = O.generalized_eigensolver_dense(A, L, "GJD", 40, 1e-8, md, B)
    lam, vec, it = fd.generalized_eigensolver(A, L, "GJD", 40, 1e-8, md, B)
    assert it == it_o
    assert np.abs(lam - lam_o).max() < EV_TOL
    assert (residuals(A, B, lam, vec) < 1e-8).all()


def test_engine_that_does_not_fit_fails_with_a_message_and_releases_memory():
    """Panels of 10^7 x 2064 doubles (5 x 165 GB) cannot be allocated: dav_create must report it, free what
    it got and leave the device usable."""
    import torch
    free0, _ = torch.cuda.mem_get_info()
    with pytest.raises(fd.DavidsonHipError, match="hipMalloc"):
        fd.CEngine(n=10_000_000, max_cols=2048)
    free1, _ = torch.cuda.mem_get_info()
    assert free1 > free0 - (1 << 30)
    lam, _, it = fd.generalized_eigensolver(O.generate_diagonal_dominant(64, 1e-2, seed=1), 2, "DPR", 50, 1e-8)
    assert it <= 50 and np.isfinite(lam).all()


@pytest.mark.parametrize("n,L,method", [(10, 3, "DPR"), (10, 3, "GJD"), (7, 2, "GJD"), (9, 4, "DPR"), (20, 10, "DPR"), (6, 3, "DPR"),
                                        (13, 5, "GJD")])
@pytest.mark.parametrize("policy", ["all", "unconverged"])
def test_basis_never_outgrows_the_space(n, L, method, policy, monkeypatch):
    """Tiny orders where doubling the basis would need more columns than the space has: the reference stops
    with an illegal-argument error in DORGQR (src/lapack_wrapper.f90:176-236); the engine completes the basis
    with the leading corrections, after which the Ritz problem is exact.  Checked against LAPACK."""
    if policy != "all":
        monkeypatch.setenv("DAVIDSON_CORRECTION_POLICY", policy)
    A = O.generate_diagonal_dominant(n, 1e-2, seed=2)
    lam, vec, it = fd.generalized_eigensolver(A, L, method, 50, 1e-8)
    ref = np.linalg.eigvalsh(A)[:L]
    assert it <= 50
    assert np.abs(lam - ref).max() < EV_TOL
    assert (residuals(A, None, lam, vec) < 1e-8).all()


def _weird_cases():
    n = 300
    rng = np.random.default_rng(0)
    cases = {"diagonal": (np.diag(np.arange(1.0, n + 1)), 4, "DPR", 60, 1e-8),
             "zero": (np.zeros((n, n)), 3, "DPR", 5, 1e-8),
             "identity": (np.eye(n), 3, "DPR", 5, 1e-8)}
    A = O.generate_diagonal_dominant(n, 1e-2, seed=5)
    d = np.arange(1.0, n + 1); d[:6] = 1.0
    A[np.arange(n), np.arange(n)] = d
    cases["degenerate_dpr"] = (A, 4, "DPR", 60, 1e-8)
    cases["degenerate_gjd"] = (A, 4, "GJD", 60, 1e-8)
    cases["negative_definite"] = (-O.generate_diagonal_dominant(n, 1e-2, seed=6), 3, "DPR", 60, 1e-8)
    base = O.generate_diagonal_dominant(n, 1e-3, seed=7)
    cases["max_it_1"] = (base, 3, "DPR", 1, 1e-8)
    cases["tol_1e-13"] = (base, 3, "DPR", 60, 1e-13)
    cases["tol_1e-2"] = (base, 3, "DPR", 60, 1e-2)
    cases["scaled_1e8_unreachable_tolerance"] = (1e8 * base, 3, "DPR", 20, 1e-8)
    cases["scaled_1e-8"] = (1e-8 * base, 3, "DPR", 60, 1e-8)
    G = rng.standard_normal((n, n))
    cases["gaussian_dpr_not_converging"] = ((G + G.T) / 2, 3, "DPR", 12, 1e-8)
    return cases


@pytest.mark.parametrize("name", sorted(_weird_cases()))
def test_unusual_inputs_behave_like_the_reference_statement(name):
    """Degenerate, indefinite, badly scaled and non-converging inputs: same iteration count and the same Ritz
    values as the oracle (which states the reference), including what is returned when nothing converges."""
    A, L, method, max_it, tol = _weird_cases()[name]
    lam_o, _, it_o = O.generalized_eigensolver_dense(A, L, method, max_it, tol)
    lam, vec, it = fd.generalized_eigensolver(A, L, method, max_it, tol)
    assert it == it_o
    scale = max(1.0, np.abs(lam_o).max())
    # not converged: two floating-point orderings of a wandering iteration drift apart, only roughly equal
    assert np.abs(lam - lam_o).max() < (1e-6 if it <= max_it else 1e-2) * scale
    if it <= max_it:                                        # converged: the bar of BASELINE.json
        assert np.abs(lam - lam_o).max() < EV_TOL * scale
        assert (residuals(A, None, lam, vec) < max(tol, 1e-13 * scale)).all()


@pytest.mark.parametrize("n,L,nranks,method,gev", [(20, 2, 3, "DPR", False), (20, 2, 4, "GJD", False), (50, 3, 4, "DPR", True),
                                                   (33, 1, 2, "DPR", False), (17, 2, 2, "GJD", True)])
def test_row_slabs_smaller_than_a_tile_and_empty_slabs(n, L, nranks, method, gev):
    """More ranks than 16-row slabs: trailing ranks own one row or none at all and still take part in every
    collective; same eigenpairs and iteration count as the single-rank solve."""
    import ctypes as C
    import threading
    A = O.generate_diagonal_dominant(n, 1e-2, seed=3)
    B = O.generate_diagonal_dominant(n, 1e-2, 1.0, seed=4) if gev else None
    lam1, vec1, it1 = fd.generalized_eigensolver(A, L, method, 60, 1e-8, None, B)
    engs = [fd.DavidsonEngine(n, L, None, gev=gev, rank=r, nranks=nranks) for r in range(nranks)]
    handles = (C.c_void_p * nranks)(*[e.c.h for e in engs])
    assert fd.hip_lib().dav_local_group_join(handles, nranks) == 0
    assert sum(e.c.local_rows()[1] for e in engs) == n and engs[-1].c.local_rows()[1] < 16
    out = [None] * nranks

    def work(r):
        engs[r].set_dense(1, A)
        if gev:
            engs[r].set_dense(2, B)
        out[r] = engs[r].solve(method, 60, 1e-8)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in threads]
    [t.join(timeout=120) for t in threads]
    assert all(o is not None for o in out), "a rank did not finish"
    for lam, vec, it in out:
        assert it == it1 and np.abs(lam - lam1).max() < 1e-12
        assert np.abs(np.abs(vec) - np.abs(vec1)).max() < 1e-10
    for e in engs:
        e.close()


def _solve_on_ranks(nranks, make, prepare, method, max_it, tol):
    """nranks DavidsonEngines as threads on one GPU (loopback transport); returns the (lam, vec, iters) of every rank"""
    import ctypes as C
    import threading
    engs = [make(r) for r in range(nranks)]
    handles = (C.c_void_p * nranks)(*[e.c.h for e in engs])
    assert fd.hip_lib().dav_local_group_join(handles, nranks) == 0
    out, err = [None] * nranks, [None] * nranks

    def work(r):
        try:
            prepare(engs[r])
            out[r] = engs[r].solve(method, max_it, tol)
        except Exception as exc:      # noqa: BLE001
            err[r] = exc
    threads = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in threads]
    [t.join(timeout=300) for t in threads]
    for e in engs:
        e.close()
    assert all(x is None for x in err), err
    assert all(o is not None for o in out)
    return out


@pytest.mark.parametrize("nranks", [2, 3])
def test_harness_operator_generated_in_the_symmetric_sweep_over_ranks(golden, nranks):
    """the reference's test operator (transcendental entries) as a device operator, symmetric generation dealt out over
    ranks: the golden values of the reference's matrix-free run"""
    manifest, arrays = golden
    case = manifest["free"]["free_n300"]

    def prepare(eng):
        eng.set_harness_operator(1)
        eng.set_harness_operator(2)
    out = _solve_on_ranks(nranks, lambda r: fd.DavidsonEngine(case["n"], case["lowest"], case["max_dim"], gev=True, rank=r, nranks=nranks,
                                                              storage="symmetric"), prepare, "DPR", case["max_it"], case["tol"])
    for lam, vec, iters in out:
        assert np.abs(lam - arrays["free_n300__evals"]).max() < 1e-8
        assert iters == case["iters"]
        assert np.array_equal(lam, out[0][0])


def test_fp32_inner_sweeps_over_ranks(golden):
    """mixed-precision GJD with the symmetric tiles (and their fp32 copies) dealt out over 2 ranks"""
    manifest, arrays = golden
    name = "n400_gev_gjd"
    case = manifest["dense"][name]
    A, B = case_matrices(case, arrays)

    def prepare(eng):
        eng.set_dense(1, A)
        eng.set_dense(2, B)
        eng.set_inner_precision(32)
    out = _solve_on_ranks(2, lambda r: fd.DavidsonEngine(case["n"], case["lowest"], case["max_dim"], gev=True, rank=r, nranks=2,
                                                         storage="symmetric"), prepare, "GJD", case["max_it"], case["tol"])
    for lam, vec, iters in out:
        assert np.abs(lam - arrays[f"{name}__evals"]).max() < EV_TOL
        assert iters == case["iters"]
        assert (residuals(A, B, lam, vec) < case["tol"]).all()


@pytest.mark.parametrize("name", ["matrix_txt_dpr", "c1_n50_gev_dpr", "n400_std_dpr", "n1000_gev_restart_dpr", "n400_gev_gjd"])
def test_drop_in_call_with_symmetric_storage_from_the_environment(golden, name, monkeypatch):
    """DAVIDSON_STORAGE=symmetric: the reference-signature dense call uploads and keeps only the lower block triangle of
    its (symmetric) input - block-column panels over PCIe, cut into tiles on the device; golden values unchanged"""
    monkeypatch.setenv("DAVIDSON_STORAGE", "symmetric")
    manifest, arrays = golden
    case = manifest["dense"][name]
    A, B = case_matrices(case, arrays)
    lam, vec, iters = fd.generalized_eigensolver(A, case["lowest"], case["method"], case["max_it"], case["tol"], case["max_dim"], B)
    assert np.abs(lam - arrays[f"{name}__evals"]).max() < EV_TOL
    assert (residuals(A, B, lam, vec) < case["tol"]).all()
    assert iters == case["iters"]
