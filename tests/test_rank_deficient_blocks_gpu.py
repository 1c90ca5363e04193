"""Correction blocks that are rank deficient BY STRUCTURE.  For a banded matrix the DPR correction t = r / (theta - d) has the
support of the residual: with the start vectors e_1 .. e_2L and bandwidth 2 all 2L corrections live in rows 1 .. 2L + 2, i.e. after
projection against the basis the block of 2L columns has rank 2 - and rounding noise cannot supply the missing directions, because
it has the same support.  The reference's Householder QR (src/davidson.f90:197-215) completes the basis with unit vectors there and
converges in 2 iterations (checked against the compiled reference: oracle/_ref, 2 iterations for every case below); this driver
completes such blocks with the unit vectors at the next entries of the start order (dav_panel_unit_column) - until round 5 it
rescaled the noise for ever and diverged."""
import numpy as np
import pytest

import fortran_davidson_amd as fd
from oracle import davidson_oracle as O

pytestmark = pytest.mark.gpu


def band(n, d0, dstep, eps):
    a = np.diag(d0 + dstep * np.arange(n, dtype=np.float64))
    for off, w in ((1, eps), (2, 0.5 * eps)):
        a += w * (np.eye(n, k=off) + np.eye(n, k=-off))
    return a


@pytest.mark.parametrize("storage", ["full", "symmetric"])
@pytest.mark.parametrize("n,lowest,method,gev", [(1000, 4, "DPR", False), (1000, 4, "DPR", True), (600, 3, "GJD", False), (600, 3, "GJD", True),
                                                 (1500, 8, "DPR", False)])
def test_banded_matrix_converges_like_the_reference(monkeypatch, storage, n, lowest, method, gev):
    monkeypatch.setenv("DAVIDSON_STORAGE", storage)
    a = band(n, 1.0, 1.0, 0.3)
    b = band(n, 1.0, 0.0, 0.05) if gev else None
    with np.errstate(invalid="ignore"):
        lam_o, _, it_o = O.generalized_eigensolver_dense(a, lowest, method, 200, 1e-8, None, b)
    assert it_o == 2                                            # = the compiled reference's count (module docstring)
    lam, vec, it = fd.generalized_eigensolver(a, lowest, method, 200, 1e-8, None, b)
    assert it == it_o
    assert np.abs(lam - lam_o).max() < 1e-10
    bx = vec if b is None else b @ vec
    assert np.linalg.norm(a @ vec - bx * lam[None, :], axis=0).max() < 1e-8


def test_block_diagonal_matrix_with_identical_blocks():
    """Two decoupled identical blocks: corrections of degenerate pairs coincide (columns that depend on EACH OTHER, not on the basis)."""
    rng = np.random.default_rng(5)
    h = rng.standard_normal((300, 300)) * 1e-2
    blk = np.diag(1.0 + np.arange(300.0)) + h + h.T
    a = np.zeros((600, 600))
    a[:300, :300] = blk
    a[300:, 300:] = blk
    lam_np = np.linalg.eigvalsh(a)[:6]
    lam, vec, it = fd.generalized_eigensolver(a, 6, "DPR", 300, 1e-8)
    assert it <= 300 and np.abs(lam - lam_np).max() < 1e-8
    assert np.linalg.norm(a @ vec - vec * lam[None, :], axis=0).max() < 1e-7


@pytest.mark.parametrize("storage,gev", [("symmetric", False), ("full", True)])
def test_banded_matrix_on_three_ranks(storage, gev):
    """The completion of a rank-deficient block is a decision every rank must take alike (all-reduced Gram blocks, the same start
    order on every rank): three ranks over the loopback transport against one rank - same iteration count, same eigenvalues."""
    import ctypes as C
    import threading
    n, lowest, nranks = 1111, 4, 3
    a = band(n, 1.0, 1.0, 0.3)
    b = band(n, 1.0, 0.0, 0.05) if gev else None

    def load(eng):
        eng.set_dense(1, a)
        if gev:
            eng.set_dense(2, b)

    with fd.DavidsonEngine(n, lowest, None, gev=gev, storage=storage) as one:
        load(one)
        lam1, _, it1 = one.solve("DPR", 200, 1e-8, want_vectors=False)
    assert it1 == 2
    engs = [fd.DavidsonEngine(n, lowest, None, gev=gev, rank=r, nranks=nranks, storage=storage) for r in range(nranks)]
    handles = (C.c_void_p * nranks)(*[e.c.h for e in engs])
    assert fd.hip_lib().dav_local_group_join(handles, nranks) == 0
    out, err = [None] * nranks, [None] * nranks

    def work(r):
        try:
            load(engs[r])
            out[r] = engs[r].solve("DPR", 200, 1e-8, want_vectors=False)
        except Exception as exc:      # noqa: BLE001
            err[r] = exc
        finally:
            fd.hip_lib().dav_local_group_yield(engs[r].c.h)

    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join() for t in th]
    for e in engs:
        e.close()
    assert all(x is None for x in err), err
    for lam, _, it in out:
        assert it == it1 and np.abs(lam - lam1).max() < 1e-10
