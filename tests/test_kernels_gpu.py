"""GPU: every HIP kernel family against numpy on the same seeded inputs, through the C ABI."""
import numpy as np
import pytest

import fortran_davidson_amd as fd
from fortran_davidson_amd.engine_c import (OP_A, OP_B, PANEL_V, PANEL_W, PANEL_BV, PANEL_X, PANEL_R, PANEL_S,
                                           METHOD_DPR, METHOD_GJD)
from oracle import davidson_oracle as O

pytestmark = pytest.mark.gpu
RTOL = 1e-13      # fp64 products of O(1) data; contraction lengths <= 3000


def relerr(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


@pytest.mark.parametrize("n", [50, 257, 1000])
def test_generated_matrix_is_bit_identical_to_oracle(n):
    with fd.CEngine(n=n, max_cols=32) as e:
        e.set_dense_generated(OP_A, seed=3, sparsity=1e-3)
        A = O.generate_diagonal_dominant(n, 1e-3, seed=3)
        assert np.array_equal(e.get_diagonal(OP_A), np.diag(A))
        # A * e_j is column j exactly (every other product is an exact zero)
        cols = [0, 1, n // 2, n - 1]
        X = np.zeros((n, len(cols)))
        X[cols, range(len(cols))] = 1.0
        e.panel_put(PANEL_V, 0, X)
        e.apply(OP_A, PANEL_V, 0, len(cols), PANEL_W, 0)
        assert np.array_equal(e.panel_get(PANEL_W, 0, len(cols)), A[:, cols])
        e.set_dense_generated(OP_B, seed=4, sparsity=1e-3, diag_val=1.0) if False else None


@pytest.mark.parametrize("n,k", [(50, 1), (50, 6), (300, 8), (300, 16), (777, 17), (1000, 32), (1000, 48),
                                 (1000, 64), (600, 100), (2500, 24)])
def test_block_matvec_dense(n, k):
    rng = np.random.default_rng(n + k)
    A = rng.standard_normal((n, n))
    A = A + A.T
    X = rng.standard_normal((n, k))
    with fd.CEngine(n=n, max_cols=max(k, 16)) as e:
        e.set_dense_host(OP_A, A)
        e.panel_put(PANEL_V, 0, X)
        e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
        W = e.panel_get(PANEL_W, 0, k)
        assert relerr(W, A @ X) < RTOL * n
        # column offsets on both sides
        if k >= 6:
            e.apply(OP_A, PANEL_V, 2, 3, PANEL_S, 1)
            assert relerr(e.panel_get(PANEL_S, 1, 3), A @ X[:, 2:5]) < RTOL * n
        st = e.stats()
        assert st.applies >= 1 and st.apply_bytes > 0


@pytest.mark.parametrize("storage", [0, 1])
@pytest.mark.parametrize("kind", ["hashed", "harness"])
def test_block_matvec_matrix_free(kind, storage):
    """storage 1 = symmetric mode: every entry of the lower block triangle is generated once, used twice."""
    n, k = 300, 9
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, k))
    with fd.CEngine(n=n, max_cols=16, gev=True) as e:
        e.set_storage(storage)
        if kind == "hashed":
            e.set_operator_hashed(OP_A, 11, 1e-3)
            e.set_operator_hashed(OP_B, 12, 1e-3, 1.0)
            A = O.generate_diagonal_dominant(n, 1e-3, seed=11)
            B = O.generate_diagonal_dominant(n, 1e-3, 1.0, seed=12)
        else:
            tab = O.harness_exp_table(n)
            e.set_operator_harness(OP_A, tab)
            e.set_operator_harness(OP_B, tab)
            A, B = O.harness_matrices(n)
        assert np.allclose(e.get_diagonal(OP_A), np.diag(A), rtol=1e-14)
        assert np.allclose(e.get_diagonal(OP_B), np.diag(B), rtol=1e-14)
        e.panel_put(PANEL_V, 0, X)
        e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
        e.apply(OP_B, PANEL_V, 0, k, PANEL_BV, 0)
        assert relerr(e.panel_get(PANEL_W, 0, k), A @ X) < 1e-12
        assert relerr(e.panel_get(PANEL_BV, 0, k), B @ X) < 1e-12


def test_identity_operator_and_edge_shapes():
    n = 40
    X = np.random.default_rng(1).standard_normal((n, 5))
    with fd.CEngine(n=n, max_cols=16, gev=True) as e:
        e.set_operator_identity(OP_B)
        e.panel_put(PANEL_V, 0, X)
        e.apply(OP_B, PANEL_V, 0, 5, PANEL_BV, 0)
        assert np.array_equal(e.panel_get(PANEL_BV, 0, 5), X)
        with pytest.raises(fd.DavidsonHipError):
            e.apply(OP_A, PANEL_V, 0, 5, PANEL_W, 0)       # operator A not set
        with pytest.raises(fd.DavidsonHipError):
            e.panel_get(PANEL_V, 0, 10_000)                # out of range


@pytest.mark.parametrize("n,p,q", [(50, 6, 6), (300, 12, 6), (1000, 16, 16), (1000, 40, 24), (2100, 128, 64),
                                   (999, 33, 17)])
def test_gram(n, p, q):
    rng = np.random.default_rng(p * q)
    P = rng.standard_normal((n, p))
    Q = rng.standard_normal((n, q))
    with fd.CEngine(n=n, max_cols=max(p, q)) as e:
        e.panel_put(PANEL_V, 0, P)
        e.panel_put(PANEL_W, 0, Q)
        G = e.gram(PANEL_V, 0, p, PANEL_W, 0, q)
        assert relerr(G, P.T @ Q) < RTOL * n
        # run-to-run reproducible (fixed-order reduction, no atomics)
        assert np.array_equal(G, e.gram(PANEL_V, 0, p, PANEL_W, 0, q))
        if p > 4 and q > 3:
            G2 = e.gram(PANEL_V, 2, p - 4, PANEL_W, 1, q - 3)
            assert relerr(G2, P[:, 2:p - 2].T @ Q[:, 1:q - 2]) < RTOL * n


@pytest.mark.parametrize("n,p,q", [(50, 6, 6), (300, 12, 3), (1000, 64, 64), (777, 30, 18), (1500, 128, 16)])
def test_panel_transform(n, p, q):
    rng = np.random.default_rng(p + q)
    P = rng.standard_normal((n, p))
    M = rng.standard_normal((p, q))
    with fd.CEngine(n=n, max_cols=max(p, q)) as e:
        e.panel_put(PANEL_V, 0, P)
        e.panel_transform(PANEL_V, 0, p, M, PANEL_X, 0)
        assert relerr(e.panel_get(PANEL_X, 0, q), P @ M) < RTOL * p
        e.panel_transform(PANEL_V, 0, p, M, PANEL_V, 0)          # in place (restart shape)
        assert relerr(e.panel_get(PANEL_V, 0, q), P @ M) < RTOL * p


@pytest.mark.parametrize("gev", [False, True])
@pytest.mark.parametrize("n,m,L", [(50, 6, 3), (400, 12, 3), (1000, 32, 8), (1300, 64, 8)])
def test_ritz_residual_dpr_phase(n, m, L, gev):
    rng = np.random.default_rng(n + m)
    A = O.generate_diagonal_dominant(n, 1e-2, seed=2)
    B = O.generate_diagonal_dominant(n, 1e-2, 1.0, seed=3) if gev else None
    V = np.linalg.qr(rng.standard_normal((n, m)))[0]
    W = A @ V
    BV = B @ V if gev else V
    H = V.T @ W
    S = V.T @ BV if gev else None
    theta, Y = O.lapack_generalized_eigensolver(H, S)
    with fd.CEngine(n=n, max_cols=2 * m, gev=gev) as e:
        e.set_dense_host(OP_A, A)
        if gev:
            e.set_dense_host(OP_B, B)
        e.panel_put(PANEL_V, 0, V)
        e.panel_put(PANEL_W, 0, W)
        if gev:
            e.panel_put(PANEL_BV, 0, BV)
        res = e.ritz_residual_correction(m, L, Y, theta, METHOD_DPR)
        X = V @ Y
        R = W @ Y - (BV @ Y) * theta[None, :]
        assert relerr(e.panel_get(PANEL_X, 0, L), X[:, :L]) < 1e-12
        assert np.allclose(res, np.linalg.norm(R[:, :L], axis=0), rtol=1e-9, atol=1e-13)
        T = O.compute_DPR_generalized_dense(A, theta, R, B)
        Tdev = e.panel_get(PANEL_V, m, m)
        assert relerr(Tdev, T) < 1e-9            # R carries cancellation; compare at residual accuracy
        # GJD mode leaves raw residues and all m Ritz vectors
        res2 = e.ritz_residual_correction(m, L, Y, theta, METHOD_GJD)
        assert np.allclose(res2, res, rtol=1e-12)
        assert relerr(e.panel_get(PANEL_R, 0, m), R) < 1e-11
        assert relerr(e.panel_get(PANEL_X, 0, m), X) < 1e-12


@pytest.mark.parametrize("n,m,kt", [(60, 6, 6), (500, 16, 16), (1200, 64, 64), (300, 0, 8)])
def test_block_gram_schmidt_phase(n, m, kt):
    rng = np.random.default_rng(m + kt)
    V = np.linalg.qr(rng.standard_normal((n, max(m, 1))))[0][:, :m]
    T = rng.standard_normal((n, kt)) + (V @ rng.standard_normal((m, kt)) if m else 0)
    with fd.CEngine(n=n, max_cols=m + kt) as e:
        if m:
            e.panel_put(PANEL_V, 0, V)
        e.panel_put(PANEL_V, m, T)
        for _ in range(2):
            Cm, G = e.ortho_gram(m, kt)
            Gp = G - Cm.T @ Cm
            w, U = np.linalg.eigh(Gp)
            M = U / np.sqrt(w)[None, :]
            e.ortho_apply(m, kt, Cm, M)
        Q = e.panel_get(PANEL_V, 0, m + kt)
        assert np.abs(Q.T @ Q - np.eye(m + kt)).max() < 1e-13
        # same span as [V, T]
        full = np.hstack([V, T])
        assert np.linalg.norm(full - Q @ (Q.T @ full)) < 1e-10 * np.linalg.norm(full)


@pytest.mark.parametrize("gev", [False, True])
@pytest.mark.parametrize("n,m,kt,nranks", [(500, 16, 16, 1), (1200, 64, 64, 1), (700, 32, 8, 1), (900, 32, 32, 3)])
def test_last_gram_schmidt_pass_fused_with_the_projection(n, m, kt, nranks, gev):
    """dav_project_ortho + dav_ortho_apply_all (round 5): the block is swept as the first pass left it, ONE fetch returns
    [V T']^T (A T'), [V T']^T (B T'), V^T T' and T'^T T'; the second pass then moves T', A T' and B T' together, and the
    projected blocks of the final T'' follow on the host.  Against the separate route (second pass, sweep of T'', dav_project)
    and against numpy."""
    rng = np.random.default_rng(n + m + kt)
    A = rng.standard_normal((n, n)); A = A + A.T
    B = rng.standard_normal((n, n)); B = B @ B.T / n + np.eye(n)
    V = np.linalg.qr(rng.standard_normal((n, m)))[0]
    T1 = rng.standard_normal((n, kt))
    T1 -= V @ (V.T @ T1)
    T1 = np.linalg.qr(T1)[0] + 1e-7 * (rng.standard_normal((n, kt)) + V @ rng.standard_normal((m, kt)))   # what a first pass leaves
    Hvv = V.T @ A @ V
    Svv = V.T @ B @ V

    def work(r, e):
        e.set_dense_host(OP_A, A)
        if gev:
            e.set_dense_host(OP_B, B)
        e.panel_put(PANEL_V, 0, V)
        e.panel_put(PANEL_V, m, T1)
        e.expand(0, m)
        # fused route
        e.expand(m, kt)
        Hraw, Sraw, C2, G2 = e.project_ortho(m, kt, gev)
        assert relerr(C2, V.T @ T1) < 1e-9 and relerr(G2, T1.T @ T1) < 1e-12
        assert relerr(Hraw, np.hstack([V, T1]).T @ (A @ T1)) < 1e-12
        Gp = G2 - C2.T @ C2
        w, U = np.linalg.eigh(Gp)
        M2 = U / np.sqrt(w)[None, :]
        e.ortho_apply_all(m, kt, C2, M2)
        T2 = (T1 - V @ C2) @ M2
        out = {}
        for name, raw, pvv, Op, panel in (("H", Hraw, Hvv, A, PANEL_W),) + ((("S", Sraw, Svv, B, PANEL_BV),) if gev else ()):
            pc = pvv @ C2
            newv = (raw[:m] - pc) @ M2
            tt = raw[m:] - C2.T @ raw[:m] - raw[:m].T @ C2 + C2.T @ pc
            newt = M2.T @ tt @ M2
            assert relerr(newv, V.T @ Op @ T2) < 1e-11, name
            assert relerr(newt, T2.T @ Op @ T2) < 1e-11, name
            assert relerr(e.panel_get(panel, m, kt), Op @ T2) < 1e-11, name       # the image followed the block
            out[name] = (newv, newt)
        Tdev = e.panel_get(PANEL_V, m, kt)
        assert relerr(Tdev, T2) < 1e-12
        Q = np.hstack([V, Tdev])
        assert np.abs(Q.T @ Q - np.eye(m + kt)).max() < 1e-12
        # separate route on the same block: sweep of T'' and dav_project
        e.expand(m, kt)
        H = np.zeros((m + kt, m + kt), order="F"); S = np.zeros((m + kt, m + kt), order="F")
        e.project(m, kt, H, S if gev else None)
        assert relerr(H[:m, m:], out["H"][0]) < 1e-11 and relerr(H[m:, m:], out["H"][1]) < 1e-11
        if gev:
            assert relerr(S[:m, m:], out["S"][0]) < 1e-11 and relerr(S[m:, m:], out["S"][1]) < 1e-11
        return True

    if nranks == 1:
        with fd.CEngine(n=n, max_cols=m + kt, gev=gev) as e:
            assert work(0, e)
    else:
        assert all(_run_ranks(nranks, lambda r: fd.CEngine(n=n, max_cols=m + kt, gev=gev, rank=r, nranks=nranks), work))


def test_init_basis_expand_project_restart():
    n, L = 500, 4
    A = O.generate_diagonal_dominant(n, 1e-2, seed=6)
    A[np.arange(n), np.arange(n)] = np.random.default_rng(0).permutation(n) + 1.0   # scrambled diagonal
    with fd.CEngine(n=n, max_cols=32) as e:
        e.set_dense_host(OP_A, A)
        idx = e.init_basis(2 * L)
        assert np.array_equal(idx - 1, O.lowest_diagonal_indices(np.diag(A), 2 * L))
        V0 = e.panel_get(PANEL_V, 0, 2 * L)
        assert np.array_equal(V0, O.generate_preconditioner(np.diag(A).copy(), 2 * L))
        assert np.array_equal(e.panel_get(PANEL_W, 0, 2 * L), A[:, idx - 1])
        H = np.zeros((32, 32), order="F")
        e.project(0, 2 * L, H)
        assert np.allclose(H[:8, :8], V0.T @ A @ V0, atol=1e-13)
        # grow by a random orthonormal block
        T = np.linalg.qr(np.random.default_rng(1).standard_normal((n, 8)))[0]
        e.panel_put(PANEL_V, 8, T)
        e.expand(8, 8)
        e.project(8, 8, H)
        V = e.panel_get(PANEL_V, 0, 16)
        assert np.allclose(H[:16, :16], V.T @ A @ V, atol=1e-12)
        assert np.allclose(e.panel_get(PANEL_W, 0, 16), A @ V, atol=1e-12)
        Y = np.linalg.qr(np.random.default_rng(2).standard_normal((16, 16)))[0]
        e.restart(16, 8, Y)
        Vn = e.panel_get(PANEL_V, 0, 8)
        assert np.allclose(Vn, V @ Y[:, :8], atol=1e-13)
        # W = A V is contracted with the same columns: no sweep of A follows a restart (src/davidson.f90:218, :223)
        assert np.allclose(e.panel_get(PANEL_W, 0, 8), A @ Vn, atol=1e-11)
        st = e.stats()
        assert st.restarts == 1 and st.m == 8 and st.applies == 1       # the one block sweep of the expansion above
        e.project(0, 8, H)
        assert np.allclose(H[:8, :8], Vn.T @ A @ Vn, atol=1e-11)


def test_restart_contracts_the_second_operator_panel_too():
    """Generalized problem: V, W = A V and B V all carry the restart transform (the Fortran driver passes the kept Ritz
    vectors times the k x k matrix that makes V Y Euclidean-orthonormal)."""
    n = 400
    A = O.generate_diagonal_dominant(n, 1e-2, seed=6)
    B = O.generate_diagonal_dominant(n, 1e-2, 1.0, seed=7)
    rng = np.random.default_rng(3)
    V = np.linalg.qr(rng.standard_normal((n, 12)))[0]
    with fd.CEngine(n=n, max_cols=32, gev=True) as e:
        e.set_dense_host(OP_A, A)
        e.set_dense_host(OP_B, B)
        e.panel_put(PANEL_V, 0, V)
        e.expand(0, 12)
        M = rng.standard_normal((12, 6))
        e.restart(12, 6, M)
        Vn = e.panel_get(PANEL_V, 0, 6)
        assert np.allclose(Vn, V @ M, atol=1e-12)
        assert np.allclose(e.panel_get(PANEL_W, 0, 6), A @ Vn, atol=1e-10)
        assert np.allclose(e.panel_get(PANEL_BV, 0, 6), B @ Vn, atol=1e-11)


def test_row_partition_matches_engine():
    from fortran_davidson_amd.distributed import RowPartition
    for n, p in [(1000, 1), (1000, 3), (50, 2), (20000, 8)]:
        for r in range(p):
            with fd.CEngine(n=n, max_cols=16, rank=r, nranks=p) as e:
                part = RowPartition(n, p, r)
                assert e.local_rows() == (part.row0, part.nloc)


@pytest.mark.parametrize("n,k", [(50, 3), (256, 16), (300, 8), (777, 17), (1000, 32), (1300, 40), (2500, 64)])
def test_block_matvec_symmetric_tiled_storage(n, k):
    """K1s: lower block triangle only in HBM, every off-diagonal tile used twice."""
    rng = np.random.default_rng(n + k)
    A = rng.standard_normal((n, n))
    A = A + A.T
    X = rng.standard_normal((n, k))
    with fd.CEngine(n=n, max_cols=max(k, 16)) as e:
        e.set_storage(1)
        e.set_dense_host(OP_A, A)
        assert np.array_equal(e.get_diagonal(OP_A), np.diag(A))
        e.panel_put(PANEL_V, 0, X)
        e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
        W = e.panel_get(PANEL_W, 0, k)
        assert relerr(W, A @ X) < RTOL * n
        # reproducible: fixed-order slab sums
        e.apply(OP_A, PANEL_V, 0, k, PANEL_S, 0)
        assert np.array_equal(W, e.panel_get(PANEL_S, 0, k))


def test_symmetric_tiled_generated_matrix_and_solver_phases():
    n, L = 700, 4
    A = O.generate_diagonal_dominant(n, 1e-2, seed=6)
    with fd.CEngine(n=n, max_cols=32) as e:
        e.set_storage(1)
        e.set_dense_generated(OP_A, 6, 1e-2)
        assert np.array_equal(e.get_diagonal(OP_A), np.diag(A))
        idx = e.init_basis(2 * L)
        assert np.array_equal(e.panel_get(PANEL_W, 0, 2 * L), A[:, idx - 1])       # column gather from tiles
        X = np.random.default_rng(0).standard_normal((n, 5))
        e.panel_put(PANEL_S, 0, X)
        e.apply(OP_A, PANEL_S, 0, 5, PANEL_R, 0)
        assert relerr(e.panel_get(PANEL_R, 0, 5), A @ X) < 1e-12


@pytest.mark.parametrize("storage", [0, 1])
def test_dense_matrix_from_device_memory(storage):
    """dav_set_dense_dev: the matrix comes from a device buffer owned by someone else (a torch tensor)."""
    import torch
    n, k = 600, 7
    rng = np.random.default_rng(2)
    A = rng.standard_normal((n, n))
    A = A + A.T
    X = rng.standard_normal((n, k))
    t = torch.from_numpy(np.ascontiguousarray(A.T)).cuda()          # row-major of A^T == column-major of A
    with fd.CEngine(n=n, max_cols=16) as e:
        e.set_storage(storage)
        e.set_dense_dev(OP_A, t.data_ptr(), n)
        del t
        torch.cuda.empty_cache()
        e.panel_put(PANEL_V, 0, X)
        e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
        assert relerr(e.panel_get(PANEL_W, 0, k), A @ X) < RTOL * n
        assert np.array_equal(e.get_diagonal(OP_A), np.diag(A))


@pytest.mark.parametrize("env", [{"DAV_SYM_PAIR": "0"}, {"DAV_SYM_RUN": "1"}, {"DAV_SYM_RUN": "7"},
                                 {"DAV_SYM_R": "2"}, {"DAV_SYM_R": "2", "DAV_SYM_WIDE": "0"}, {"DAV_SYM_R": "2", "DAV_SYM_WIDE": "1"},
                                 {"DAV_SYM_R": "2", "DAV_SYM_WIDE": "2", "DAV_SYM_RUN9": "1"}, {"DAV_SYM_R": "4"}, {"DAV_SYM_R": "4", "DAV_SYM_RUN9": "1"},
                                 {"DAV_SYM_R": "2", "DAV_SYM_RUN9": "3"}, {"DAV_SYM_R": "2", "DAV_SYM_PAIR": "0"},
                                 {"DAV_SYM_R": "2", "DAV_SYM_QUAD": "0"}, {"DAV_SYM_R": "4", "DAV_SYM_MFMA4": "0"}])
def test_symmetric_sweep_alternative_kernels_and_schedules(env, monkeypatch):
    """The A/B knobs of the symmetric sweep (one-wave-per-SIMD kernel, unpaired 16-column launches, other run
    lengths, the super-row schedules with 2 / 4 block rows per workgroup that large matrices select by
    themselves - with 2 block rows: the wide one-wave-per-SIMD kernel of k_matvec_symw.hip for more than 8 columns
    (DAV_SYM_WIDE = 2, default), for more than 16 only (1) or never (0: matvec_sym9_kernel<2>)) are read ONCE per engine, when it
    is created (csrc/engine.hip: tune_from_env) - so each set is put into the environment around the engines of this test (until
    round 5 each ran in a child process: 13 interpreter starts); same product, bit-reproducible.
    Orders cover 1..10 block rows: ragged super rows, diagonal super blocks, a single block row."""
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    for n, k in [(300, 8), (1300, 40), (2500, 64), (200, 3), (1300, 5), (1800, 8), (2305, 7), (2500, 16), (1030, 24)]:
        A, X, ref = _alt_case(n, k)
        with fd.CEngine(n=n, max_cols=max(k, 16)) as e:
            e.set_storage(1)
            e.set_dense_host(OP_A, A)
            e.panel_put(PANEL_V, 0, X)
            e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
            W = e.panel_get(PANEL_W, 0, k)
            assert np.abs(W - ref).max() <= 1e-12 * n * np.abs(ref).max(), (n, k)
            e.apply(OP_A, PANEL_V, 0, k, PANEL_S, 0)
            assert np.array_equal(W, e.panel_get(PANEL_S, 0, k))
    # the hashed operator generated in the sweep (every symmetric pair once) under the same schedule
    for n, k in [(700, 8), (1027, 17), (2500, 40), (1500, 4)]:
        X, ref = _alt_hashed_case(n, k)
        with fd.CEngine(n=n, max_cols=max(k, 16)) as e:
            e.set_storage(1)
            e.set_operator_hashed(OP_A, 13, 1e-2)
            e.panel_put(PANEL_V, 0, X)
            e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
            assert np.abs(e.panel_get(PANEL_W, 0, k) - ref).max() <= 1e-12 * np.abs(ref).max(), (n, k)


_ALT = {}


def _alt_case(n, k):
    """inputs and the numpy product of one (order, columns) case of the test above: made once, shared by its 13 knob sets"""
    if (n, k) not in _ALT:
        rng = np.random.default_rng(n + k)
        A = rng.standard_normal((n, n))
        A = A + A.T
        X = rng.standard_normal((n, k))
        _ALT[(n, k)] = (A, X, A @ X)
    return _ALT[(n, k)]


def _alt_hashed_case(n, k):
    if ("hashed", n, k) not in _ALT:
        A = O.generate_diagonal_dominant(n, 1e-2, seed=13)
        X = np.random.default_rng(n).standard_normal((n, k))
        _ALT[("hashed", n, k)] = (X, A @ X)
    return _ALT[("hashed", n, k)]


@pytest.mark.parametrize("n,k", [(50, 3), (256, 16), (700, 8), (1027, 17), (2500, 40)])
def test_hashed_operator_symmetric_generation_equals_the_dense_generator(n, k):
    """Matrix-free hashed operator in storage mode "symmetric": every entry of the lower block triangle is
    generated once and used for both products - same result as the stored matrix with the same entries."""
    A = O.generate_diagonal_dominant(n, 1e-2, seed=13)
    Au = O.generate_diagonal_dominant(n, 1e-2, 1.0, seed=13)          # unit-diagonal variant (the B operator)
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, k))
    with fd.CEngine(n=n, max_cols=max(k, 16), gev=True) as e:
        e.set_storage(1)
        e.set_operator_hashed(OP_A, 13, 1e-2)
        e.set_operator_hashed(OP_B, 13, 1e-2, 1.0)
        assert np.array_equal(e.get_diagonal(OP_A), np.diag(A))
        e.panel_put(PANEL_V, 0, X)
        for op, M in ((OP_A, A), (OP_B, Au)):
            e.apply(op, PANEL_V, 0, k, PANEL_W, 0)
            W = e.panel_get(PANEL_W, 0, k)
            ref = M @ X
            assert np.abs(W - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    # and the row-slab generation of the same operator agrees to rounding
    with fd.CEngine(n=n, max_cols=max(k, 16)) as e:
        e.set_operator_hashed(OP_A, 13, 1e-2)
        e.panel_put(PANEL_V, 0, X)
        e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
        assert np.abs(e.panel_get(PANEL_W, 0, k) - A @ X).max() <= 1e-12 * np.abs(A @ X).max()


@pytest.mark.parametrize("percent", ["0", "1", "30", "60", "85"])
@pytest.mark.parametrize("sched", ["0", "2", "4"])
def test_generated_second_operator_kept_partly_resident(percent, sched, monkeypatch):
    """configs[3]'s second operator (the generator with unit diagonal, never stored in full) with the tiles of its longest block
    rows kept resident (DAV_B_RESIDENT = percentage of the tiles at most; 1 = what the free memory allows: everything at these
    sizes; 0 = nothing): the resident block rows run the stored kernels, the others are generated, the two parts are summed in
    fixed order - same product as the dense generator's matrix, bit-reproducible, for every schedule and ragged orders."""
    monkeypatch.setenv("DAV_B_RESIDENT", percent)
    if sched != "0":
        monkeypatch.setenv("DAV_SYM_R", sched)
    for n, k in [(2500, 16), (2500, 40), (1300, 8), (3333, 64), (3333, 5), (700, 24)]:
        Bm = O.generate_diagonal_dominant(n, 1e-2, 1.0, seed=21)
        X = np.random.default_rng(n + k).standard_normal((n, k))
        with fd.CEngine(n=n, max_cols=max(k, 16), gev=True) as e:
            e.set_storage(1)
            e.set_operator_hashed(OP_B, 21, 1e-2, 1.0)
            assert np.array_equal(e.get_diagonal(OP_B), np.diag(Bm))
            e.panel_put(PANEL_V, 0, X)
            e.apply(OP_B, PANEL_V, 0, k, PANEL_W, 0)
            W = e.panel_get(PANEL_W, 0, k)
            ref = Bm @ X
            frac = e.resident_fraction(OP_B)
            assert np.abs(W - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), (n, k, frac)
            e.apply(OP_B, PANEL_V, 0, k, PANEL_S, 0)
            assert np.array_equal(W, e.panel_get(PANEL_S, 0, k))
            if percent == "0":
                assert frac == 0.0
            elif percent == "1":
                assert frac == 1.0
            else:
                assert frac <= int(percent) / 100.0 + 1e-12
                if n >= 2500 and int(percent) >= 60:
                    assert 0.0 < frac < 1.0                       # a real split: both parts ran
            # a new definition of the operator drops what was resident
            e.set_operator_identity(OP_B)
            assert e.resident_fraction(OP_B) == 0.0


@pytest.mark.parametrize("nranks", [2, 3])
def test_partly_resident_operator_over_several_ranks(nranks, monkeypatch):
    """the same split on every rank of a multi-rank engine (each rank keeps resident the longest of ITS block rows)"""
    monkeypatch.setenv("DAV_B_RESIDENT", "85")
    n, k = 3333, 24
    Bm = O.generate_diagonal_dominant(n, 1e-2, 1.0, seed=21)
    X = np.random.default_rng(3).standard_normal((n, k))

    def work(r, e):
        e.set_storage(1)
        e.set_operator_hashed(OP_B, 21, 1e-2, 1.0)
        e.panel_put(PANEL_V, 0, X)
        e.apply(OP_B, PANEL_V, 0, k, PANEL_W, 0)
        return e.panel_get(PANEL_W, 0, k), e.resident_fraction(OP_B)

    out = _run_ranks(nranks, lambda r: fd.CEngine(n=n, max_cols=32, gev=True, rank=r, nranks=nranks), work)
    ref = Bm @ X
    for W, frac in out:
        assert np.abs(W - ref).max() <= 1e-12 * np.abs(ref).max()
        assert np.array_equal(W, out[0][0])
    assert any(0.0 < frac < 1.0 for _, frac in out)


@pytest.mark.parametrize("n,k", [(51700, 8), (51700, 16), (51700, 40), (51700, 64), (16700, 16), (16700, 40), (16700, 8)])
def test_symmetric_super_row_schedules_at_a_size_that_selects_them(n, k):
    """From 200 block rows on the sweep runs the super-row schedules (4 block rows per workgroup for k <= 8, else 2)
    by itself: N=51700 (202 block rows, a ragged last super row), the same generated matrix in full storage
    as the reference, stored tiles and the hashed operator generated in the sweep.  Stored fp64 tiles and more than 8
    columns: two block rows per workgroup (the wide kernel; 9-16 columns from 200 block rows on: four) from 64 block rows on -
    N=16700 has 66, an odd last super row (the generated operator stays on the one-block-row kernel there)."""
    X = np.random.default_rng(k).standard_normal((n, k))
    with fd.CEngine(n=n, max_cols=64) as e:
        e.set_dense_generated(OP_A, 5, 1e-3)
        e.panel_put(PANEL_V, 0, X)
        e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
        ref = e.panel_get(PANEL_W, 0, k)
    for generated in (False, True):
        with fd.CEngine(n=n, max_cols=64) as e:
            e.set_storage(1)
            if generated:
                e.set_operator_hashed(OP_A, 5, 1e-3)
            else:
                e.set_dense_generated(OP_A, 5, 1e-3)
            e.panel_put(PANEL_V, 0, X)
            e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
            W = e.panel_get(PANEL_W, 0, k)
            assert np.abs(W - ref).max() <= 1e-12 * np.abs(ref).max()
            e.apply(OP_A, PANEL_V, 0, k, PANEL_S, 0)
            assert np.array_equal(W, e.panel_get(PANEL_S, 0, k))


@pytest.mark.parametrize("gen_wide", ["1", "0"])
def test_hashed_operator_wider_than_16_columns_generates_each_entry_once_per_32(gen_wide, monkeypatch):
    """The matrix-free hashed operator in symmetric generation at 17-64 columns: `matvec_symw_kernel<2, GEN>` generates a 32 x 16
    sub-block per half-step and feeds both 16-column groups from it (DAV_SYM_GEN_WIDE=0: the 16-column kernel per group, as before).
    Two-block-row schedule forced at small orders; ragged orders (last block row and last tile column partly outside the matrix),
    diagonal = i (A) and = 1 (B), block widths that leave a ragged second group; against the oracle's matrix."""
    monkeypatch.setenv("DAV_SYM_R", "2")
    monkeypatch.setenv("DAV_SYM_GEN_WIDE", gen_wide)
    for n, k in [(300, 17), (700, 32), (1300, 40), (2305, 64), (2560, 24)]:
        rng = np.random.default_rng(n + k)
        G = O.generate_diagonal_dominant(n, 1e-2, seed=21)
        GB = O.generate_diagonal_dominant(n, 1e-2, 1.0, seed=22)
        X = rng.standard_normal((n, k))
        with fd.CEngine(n=n, max_cols=64, gev=True) as e:
            e.set_storage(1)
            e.set_operator_hashed(OP_A, 21, 1e-2)
            e.set_operator_hashed(OP_B, 22, 1e-2, 1.0)
            e.panel_put(PANEL_V, 0, X)
            e.reset_stats()
            e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
            assert e.stats().apply_launches == ((k + 31) // 32 if gen_wide == "1" else (k + 15) // 16)
            e.apply(OP_B, PANEL_V, 0, k, PANEL_BV, 0)
            W, BV = e.panel_get(PANEL_W, 0, k), e.panel_get(PANEL_BV, 0, k)
            assert relerr(W, G @ X) < 1e-12, (n, k)
            assert relerr(BV, GB @ X) < 1e-12, (n, k)
            e.apply(OP_A, PANEL_V, 0, k, PANEL_S, 0)
            assert np.array_equal(W, e.panel_get(PANEL_S, 0, k))          # bitwise reproducible


def _run_ranks(nranks, make, work):
    """nranks engines as threads of this process on one GPU, collectives through the loopback transport"""
    import ctypes as C
    import threading
    engs = [make(r) for r in range(nranks)]
    handles = (C.c_void_p * nranks)(*[e.h for e in engs])
    assert fd.hip_lib().dav_local_group_join(handles, nranks) == 0
    out, err = [None] * nranks, [None] * nranks

    def run(r):
        try:
            out[r] = work(r, engs[r])
        except Exception as exc:      # noqa: BLE001
            err[r] = exc
    threads = [threading.Thread(target=run, args=(r,)) for r in range(nranks)]
    [t.start() for t in threads]
    [t.join(timeout=300) for t in threads]
    for e in engs:
        e.close()
    assert all(x is None for x in err), err
    assert all(o is not None for o in out), "a rank did not finish"
    return out


@pytest.mark.parametrize("nranks", [2, 3, 5])
@pytest.mark.parametrize("sched", ["1", "2", "4"])
def test_symmetric_sweep_over_several_ranks(nranks, sched, monkeypatch):
    """Symmetric-tiled storage dealt out over ranks by groups of 4 block rows: every rank stores and sweeps only its
    block rows against the all-gathered block, one reduce-scatter sums the partial products into row slabs.
    Stored tiles (host upload and device generator), the hashed operator generated in the sweep, every schedule,
    ragged orders, more ranks than groups of block rows (a rank without a single tile)."""
    monkeypatch.setenv("DAV_SYM_R", sched)
    for n, k in [(300, 8), (1300, 5), (2305, 16), (2500, 40), (700, 3)]:
        rng = np.random.default_rng(n + k)
        A = rng.standard_normal((n, n)); A = A + A.T
        G = O.generate_diagonal_dominant(n, 1e-2, seed=13)
        X = rng.standard_normal((n, k))

        def work(r, e):
            res = []
            e.set_storage(1)
            e.set_dense_host(OP_A, A)
            assert np.array_equal(e.get_diagonal(OP_A), np.diag(A))
            e.panel_put(PANEL_V, 0, X)
            e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
            res.append(e.panel_get(PANEL_W, 0, k))
            e.set_dense_generated(OP_A, 13, 1e-2)
            assert np.array_equal(e.get_diagonal(OP_A), np.diag(G))
            e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
            res.append(e.panel_get(PANEL_W, 0, k))
            e.set_operator_hashed(OP_A, 13, 1e-2)
            e.apply(OP_A, PANEL_V, 0, k, PANEL_S, 0)
            res.append(e.panel_get(PANEL_S, 0, k))
            return res

        out = _run_ranks(nranks, lambda r: fd.CEngine(n=n, max_cols=max(k, 16), rank=r, nranks=nranks), work)
        for res in out:
            assert relerr(res[0], A @ X) < RTOL * n
            assert relerr(res[1], G @ X) < 1e-12
            assert relerr(res[2], G @ X) < 1e-12
            for a, b in zip(res, out[0]):
                assert np.array_equal(a, b)              # every rank gathers the same bits


@pytest.mark.parametrize("nranks", [1, 2, 3, 5])
def test_first_block_of_the_basis_is_gathered_not_swept(nranks):
    """dav_init_basis: W0 = A V0 (and B V0) for the unit columns V0 = e_p of the lowest diagonal entries are COLUMNS of the operator
    (src/davidson.f90:128-135 applies the matrix to them).  Symmetric tiles dealt out over several ranks: every rank contributes what
    its tiles hold of those columns, one reduce-scatter (40 columns: two rounds of it); generated operators: the columns are generated.
    Bit for bit the operator's entries, no sweep counted; ragged orders, a scrambled diagonal, more ranks than groups of block rows."""
    for n, ncols in [(300, 6), (1300, 16), (2305, 40), (700, 33)]:
        rng = np.random.default_rng(n)
        A = rng.standard_normal((n, n)); A = A + A.T
        A[np.arange(n), np.arange(n)] = rng.permutation(n) + 1.0
        G = O.generate_diagonal_dominant(n, 1e-2, seed=13)
        GB = O.generate_diagonal_dominant(n, 1e-2, 1.0, seed=14)

        def work(r, e):
            res = []
            e.set_storage(1)
            e.set_dense_host(OP_A, A)
            e.set_operator_hashed(OP_B, 14, 1e-2, 1.0)
            e.reset_stats()
            idx = e.init_basis(ncols)
            assert e.stats().applies == 0
            res += [idx, e.panel_get(PANEL_W, 0, ncols), e.panel_get(PANEL_BV, 0, ncols)]
            e.set_operator_hashed(OP_A, 13, 1e-2)
            e.set_operator_identity(OP_B)
            idx2 = e.init_basis(ncols)
            res += [idx2, e.panel_get(PANEL_W, 0, ncols), e.panel_get(PANEL_BV, 0, ncols), e.panel_get(PANEL_V, 0, ncols)]
            return res

        out = _run_ranks(nranks, lambda r: fd.CEngine(n=n, max_cols=max(ncols, 16), gev=True, rank=r, nranks=nranks), work)
        for idx, W, BV, idx2, W2, BV2, V2 in out:
            assert np.array_equal(idx - 1, O.lowest_diagonal_indices(np.diag(A), ncols))
            assert np.array_equal(W, A[:, idx - 1])
            assert np.array_equal(BV, GB[:, idx - 1])
            assert np.array_equal(idx2, np.arange(1, ncols + 1))
            assert np.array_equal(W2, G[:, :ncols])
            assert np.array_equal(BV2, V2) and np.array_equal(V2, np.eye(n)[:, :ncols])


@pytest.mark.parametrize("nranks", [2, 3])
def test_overlapped_pipeline_of_the_symmetric_sweep_with_several_ranks(nranks, monkeypatch):
    """The chunked pipeline that several GPUs run by default for blocks wider than 32 columns (all-gather of chunk i + 1 and
    reduce-scatter of chunk i - 1 around the sweep of chunk i; Xt groups, partial-product and receive buffers alternating with
    the chunk parity) with SEVERAL ranks: over the loopback transport its collectives run on the engine's stream, everything
    else - layouts, counts, parities, the rows every rank receives - is what the RCCL run executes.  Same product as A X, the
    same bits as the serial path of the same ranks, ragged orders and widths, a rank without tiles."""
    monkeypatch.setenv("DAV_SYM_R", "2")
    out = {}
    for overlap in ("1", "0"):
        monkeypatch.setenv("DAV_SYM_OVERLAP", overlap)
        for n, k in [(2500, 40), (2305, 64), (1300, 33), (700, 48)]:
            rng = np.random.default_rng(n + k)
            A = rng.standard_normal((n, n)); A = A + A.T
            X = rng.standard_normal((n, k))

            def work(r, e):
                e.set_storage(1)
                e.set_dense_host(OP_A, A)
                e.panel_put(PANEL_V, 0, X)
                e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
                W = e.panel_get(PANEL_W, 0, k)
                e.apply(OP_A, PANEL_V, 0, k, PANEL_S, 0)
                assert np.array_equal(W, e.panel_get(PANEL_S, 0, k))
                return W

            res = _run_ranks(nranks, lambda r: fd.CEngine(n=n, max_cols=64, rank=r, nranks=nranks), work)
            for W in res:
                assert relerr(W, A @ X) < RTOL * n
                assert np.array_equal(W, res[0])
            out[(overlap, n, k)] = res[0]
    for (overlap, n, k), W in out.items():
        if overlap == "1":
            assert np.array_equal(W, out[("0", n, k)]), (n, k)       # the pipeline changes the order of launches, not a single sum


def test_ranks_that_disagree_on_a_control_decision_stop_with_a_message():
    """dav_ranks_agree: what the multi-rank driver loop calls once per iteration with its decisions; identical words pass,
    different words are an error ON EVERY RANK (nobody is left alone in the next collective)."""
    nranks = 3

    def work(r, e):
        e.ranks_agree([4.0, 32.0, 0.0, 1.0])
        try:
            e.ranks_agree([4.0, 32.0, 1.0 if r == 1 else 0.0, 1.0])
        except fd.DavidsonHipError as exc:
            return str(exc)
        return "no error"

    out = _run_ranks(nranks, lambda r: fd.CEngine(n=500, max_cols=16, rank=r, nranks=nranks), work)
    assert all("ranks disagree" in o for o in out), out


def test_solve_inputs_are_verified_in_a_fixed_size_collective_when_they_change():
    """dav_agree_inputs (round 6; round-5 advisor): the per-iteration control words ride on all-reduces whose element count follows
    from the basis width, so ranks that differ in `lowest` or `max_dim_sub` would enter those with different counts.  The INPUTS of a
    solve are therefore verified in a collective of fixed size - at the first solve of an engine and whenever they change; a repeated
    solve with the same inputs adds no collective; inputs that differ between the ranks are an error ON EVERY RANK."""
    nranks = 3

    def work(r, e):
        e.reset_stats()
        e.agree_inputs([500.0, 3.0, 30.0, 100.0, 1e-8, 0.0, 0.0, 1.0, 0.0])
        assert e.stats().collectives == 1                        # the first solve of an engine always verifies
        e.agree_inputs([500.0, 3.0, 30.0, 100.0, 1e-8, 0.0, 0.0, 1.0, 0.0])
        assert e.stats().collectives == 1                        # unchanged inputs: no collective
        e.agree_inputs([500.0, 4.0, 40.0, 100.0, 1e-8, 0.0, 0.0, 1.0, 0.0])
        assert e.stats().collectives == 2                        # every rank changed them together: verified again
        try:
            e.agree_inputs([500.0, 4.0 if r != 2 else 5.0, 41.0, 100.0, 1e-8, 0.0, 0.0, 1.0, 0.0])      # rank 2 asks for other `lowest`
        except fd.DavidsonHipError as exc:
            return str(exc)
        return "no error"

    out = _run_ranks(nranks, lambda r: fd.CEngine(n=500, max_cols=16, rank=r, nranks=nranks), work)
    assert all("ranks disagree" in o for o in out), out


def test_control_words_ride_on_the_next_all_reduced_result():
    """dav_agree_next (round 5): the driver's control words wait in the engine and ride on the next all-reduced small result - no
    collective of their own (the count shows it); identical words pass, different words fail the fetch ON EVERY RANK."""
    nranks, n = 3, 700
    rng = np.random.default_rng(3)
    X = rng.standard_normal((n, 8))

    def work(r, e):
        e.panel_put(PANEL_V, 0, X)
        e.reset_stats()
        e.agree_next([2.0, 16.0, 1.0, 1e-8])
        G = e.gram(PANEL_V, 0, 8, PANEL_V, 0, 8)                 # the words travel with this result
        assert relerr(G, X.T @ X) < 1e-12
        assert e.stats().collectives == 1
        G2 = e.gram(PANEL_V, 0, 8, PANEL_V, 0, 8)                # nothing pending: a plain fetch
        assert np.array_equal(G, G2) and e.stats().collectives == 2
        e.agree_next([2.0, 16.0, 1.0 if r != 1 else 0.0, 1e-8])
        try:
            e.gram(PANEL_V, 0, 8, PANEL_V, 0, 8)
        except fd.DavidsonHipError as exc:
            return str(exc)
        return "no error"

    out = _run_ranks(nranks, lambda r: fd.CEngine(n=n, max_cols=16, rank=r, nranks=nranks), work)
    assert all("ranks disagree" in o for o in out), out


def test_symmetric_sweep_with_collectives_overlapped_on_a_second_stream():
    """RCCL path of the multi-rank symmetric sweep for blocks wider than 32 columns with DAV_SYM_OVERLAP=1 (opt-in): the
    all-gather of the next 32 columns and the reduce-scatter of the previous ones run on a second stream under the sweep of
    the current ones (Xt column groups and partial-product buffers alternate).  Runs here through a 1-rank RCCL communicator
    (DAVIDSON_FORCE_RCCL=1) with the two-block-row schedule forced; the knob is read once per process, hence a child."""
    import os
    import subprocess
    import sys
    code = r"""
import numpy as np
import fortran_davidson_amd as fd
from fortran_davidson_amd.engine_c import OP_A, PANEL_V, PANEL_W, PANEL_S
for n, k in [(1300, 40), (2500, 64), (2305, 33), (700, 48)]:
    rng = np.random.default_rng(n + k)
    A = rng.standard_normal((n, n)); A = A + A.T
    X = rng.standard_normal((n, k))
    with fd.CEngine(n=n, max_cols=max(k, 16)) as e:
        e.comm_init(fd.CEngine.comm_unique_id())
        e.set_storage(1)
        e.set_dense_host(OP_A, A)
        e.set_timing(2)
        e.reset_stats()
        e.panel_put(PANEL_V, 0, X)
        e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
        W = e.panel_get(PANEL_W, 0, k)
        ref = A @ X
        assert np.abs(W - ref).max() <= 1e-12 * n * np.abs(ref).max(), (n, k)
        e.apply(OP_A, PANEL_V, 0, k, PANEL_S, 0)
        assert np.array_equal(W, e.panel_get(PANEL_S, 0, k))
        st = e.stats()
        assert st.applies == 2 * ((k + 31) // 32) and st.apply_cols == 2 * k
        # the collectives of this path run (and are timed, level 2) on the second stream: one grouped all-gather and one grouped
        # reduce-scatter per chunk of 32 columns
        assert st.collectives >= 2 * 2 * ((k + 31) // 32) and st.allgather_ms > 0.0 and st.reduce_scatter_ms > 0.0
        assert st.comm_overlap == 1 and st.comm_ranks == 1
print("OK")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root,
                         env=dict(os.environ, PYTHONPATH=root, DAVIDSON_FORCE_RCCL="1", DAV_SYM_R="2", DAV_SYM_OVERLAP="1"))
    assert res.returncode == 0 and "OK" in res.stdout, (res.stdout + res.stderr)[-2000:]


def test_stream_microbenchmark_reports_a_plausible_hbm_rate():
    """dav_bench_stream: copy and triad rates of the box (what bench.py quotes HBM fractions against besides the 8 TB/s spec)."""
    with fd.CEngine(n=1024, max_cols=16) as e:
        copy, triad = e.bench_stream(1 << 26, 3)       # 512 MiB per array: beyond the 256 MiB Infinity Cache
        assert 1000.0 < copy < 8000.0 and 1000.0 < triad < 8000.0, (copy, triad)
