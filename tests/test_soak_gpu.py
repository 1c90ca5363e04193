"""Repeated solves on one engine and repeated engine life cycles (profiles/tools/soak.py in small): results bitwise identical
from solve to solve (every reduction of the engine is in fixed order) and no device memory left behind - the reference allocates
and frees all work arrays inside the call (src/davidson.f90:94-96,238-244); the drop-in entry creates and destroys its engine
the same way."""
import numpy as np
import pytest

import fortran_davidson_amd as fd

pytestmark = pytest.mark.gpu


def _free_mb():
    """free device memory once the library has handed back what its buffer cache keeps of destroyed engines (dav_free_buffers)"""
    import torch
    fd.free_buffers()
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20


@pytest.mark.parametrize("gev,method,n,lowest", [(False, "DPR", 12000, 8), (True, "GJD", 4000, 4), (False, "GJD", 4000, 8)])
def test_repeated_solves_are_bitwise_identical_and_leave_no_memory_behind(gev, method, n, lowest):
    with fd.DavidsonEngine(n, lowest, None, gev=gev, storage="symmetric") as eng:
        eng.generate_diagonal_dominant(1, 1e-3, seed=1)
        if gev:
            eng.set_hashed_operator(2, 1e-3, 1.0, seed=2)
        lam0, _, it0 = eng.solve(method, 1000, 1e-8, want_vectors=False)
        f0 = _free_mb()
        for _ in range(40):
            lam, _, it = eng.solve(method, 1000, 1e-8, want_vectors=False)
            assert it == it0 and np.array_equal(lam, lam0)
        assert abs(_free_mb() - f0) < 64


def test_engine_life_cycles_of_the_drop_in_entry_leave_no_memory_behind():
    n = 1200
    A = np.asfortranarray(np.random.default_rng(0).standard_normal((n, n)))
    A = A + A.T + np.diag(np.arange(n) * 10.0)
    fd.generalized_eigensolver(A, 4, "DPR", 200, 1e-8)
    f0 = _free_mb()
    out = [fd.generalized_eigensolver(A, 4, "DPR", 200, 1e-8) for _ in range(20)]
    assert all(np.array_equal(o[0], out[0][0]) and o[2] == out[0][2] for o in out)
    assert abs(_free_mb() - f0) < 64


def test_a_matrix_that_does_not_fit_is_refused_with_a_message_and_nothing_is_left_behind():
    """An operator larger than the HBM: the C ABI returns an error (the Fortran front ends print it and stop, as the reference does for
    a failed LAPACK call, src/lapack_wrapper.f90:395-408), no memory stays allocated and the next engine works."""
    from fortran_davidson_amd.engine_c import OP_A
    f0 = _free_mb()
    for n, storage in ((300000, 1), (250000, 0)):               # 344 GB of tiles / 477 GB of rows
        with pytest.raises(Exception, match="out of memory|failed"):
            with fd.CEngine(n=n, max_cols=32) as e:
                e.set_storage(storage)
                e.set_dense_generated(OP_A, 1, 1e-3)
    assert abs(_free_mb() - f0) < 64
    with fd.DavidsonEngine(4000, 4) as eng:
        eng.generate_diagonal_dominant(1, 1e-3, seed=1)
        assert eng.solve("DPR", 100, 1e-8, want_vectors=False)[2] == 3


def test_the_dense_front_end_switches_to_symmetric_tiles_when_the_full_rows_do_not_fit():
    """generalized_eigensolver(matrix, ...) keeps the matrix as the reference's DGEMM reads it - full rows - unless they do not fit the
    device next to the panels: then only the lower block triangle is uploaded (same results; the reference assumes a symmetric
    matrix, src/davidson.f90:75-76).  The decision (fits_as_full_rows, from dav_device_memory) at orders one cannot allocate on a
    test box's host; the tiled path itself through DAVIDSON_STORAGE=symmetric."""
    import ctypes as C
    with fd.DavidsonEngine(1000, 4) as eng:
        free, total = eng.c.device_memory()
        assert 0 < free <= total and total > 200 * 2**30              # an MI355X: 288 GB
        fits = lambda n, nmat: eng.lib.fd_engine_fits_as_full_rows(eng.p, C.c_int(n), C.c_int(nmat))   # noqa: E731
        assert fits(20000, 1) == 1 and fits(100000, 2) == 1           # 3.2 GB; 2 x 80 GB
        assert fits(200000, 1) == 0 and fits(150000, 2) == 0          # 320 GB; 2 x 180 GB
        n_edge = int((0.9 * free / 8.0) ** 0.5)
        assert fits(n_edge - 50, 1) == 1 and fits(n_edge + 50, 1) == 0
    n = 1500
    A = O_matrix(n)
    lam_full, vec_full, it_full = fd.generalized_eigensolver(A, 4, "DPR", 200, 1e-8)
    import os
    os.environ["DAVIDSON_STORAGE"] = "symmetric"
    try:
        lam_sym, vec_sym, it_sym = fd.generalized_eigensolver(A, 4, "DPR", 200, 1e-8)
    finally:
        del os.environ["DAVIDSON_STORAGE"]
    assert it_sym == it_full and np.abs(lam_sym - lam_full).max() < 1e-12


def O_matrix(n):
    from oracle import davidson_oracle as O
    return O.generate_diagonal_dominant(n, 1e-3, seed=3)
