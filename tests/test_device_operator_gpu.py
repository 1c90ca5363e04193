"""The caller's own operator as a block apply on device memory (dav_set_operator_device, ABI 106): the device counterpart of the
reference's matrix-free interface (src/davidson.f90:277-337: a procedure X(N,k) -> (N,k)).  The "user" here is
tests/helpers/user_operator.hip - a banded stencil as its own HIP kernel, built into lib/test/libuser_operator.so by build().
Checked against the oracle's dense solve of the same matrix: eigenvalues, residuals AND iteration counts, on one rank and on
three ranks (the engine all-gathers the block for the callback)."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import fortran_davidson_amd as fd
from oracle import davidson_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def user():
    fd.hip_lib()            # first: one HIP runtime per process (with PyTorch present the engine binds torch's copy; the helper follows it)
    lib = C.CDLL(os.path.join(ROOT, "fortran_davidson_amd", "lib", "test", "libuser_operator.so"))
    lib.user_op_create.restype = C.c_void_p
    lib.user_op_create.argtypes = [C.c_double, C.c_double, C.c_double]
    lib.user_op_destroy.argtypes = [C.c_void_p]
    return lib


def stencil_matrix(n, d0, dstep, eps):
    a = np.diag(d0 + dstep * np.arange(n, dtype=np.float64))
    for off, w in ((1, eps), (2, 0.5 * eps)):
        a += w * (np.eye(n, k=off) + np.eye(n, k=-off))
    return a


def set_ops(eng, user, gev, keep):
    n = eng.n
    ctx_a = user.user_op_create(1.0, 1.0, 0.3)
    keep.append(ctx_a)
    eng.set_device_operator(1, user.user_op_apply, ctx_a, 1.0 + np.arange(n, dtype=np.float64))
    if gev:
        ctx_b = user.user_op_create(1.0, 0.0, 0.05)
        keep.append(ctx_b)
        eng.set_device_operator(2, user.user_op_apply, ctx_b, np.ones(n))


@pytest.mark.parametrize("gev,method,n,lowest", [(False, "DPR", 3000, 4), (True, "DPR", 2500, 3), (False, "GJD", 1500, 4), (True, "GJD", 1200, 3)])
def test_user_kernel_as_operator_matches_the_oracle_on_the_same_matrix(user, gev, method, n, lowest):
    a = stencil_matrix(n, 1.0, 1.0, 0.3)
    b = stencil_matrix(n, 1.0, 0.0, 0.05) if gev else None
    lam_o, _, it_o = O.generalized_eigensolver_dense(a, lowest, method, 200, 1e-8, None, b)
    keep = []
    with fd.DavidsonEngine(n, lowest, None, gev=gev) as eng:
        set_ops(eng, user, gev, keep)
        lam, vec, it = eng.solve(method, 200, 1e-8)
    for ctx in keep:
        user.user_op_destroy(ctx)
    assert it == it_o
    assert np.abs(lam - lam_o).max() < 1e-9
    bx = vec if b is None else b @ vec
    assert np.linalg.norm(a @ vec - bx * lam[None, :], axis=0).max() < 1e-8


@pytest.mark.parametrize("gev", [False, True])
def test_user_kernel_on_three_ranks_sees_the_gathered_block(user, gev):
    """n = 2999 over three ranks (slabs of 1008 / 1008 / 983 rows): the stencil reads across the slab boundaries, so the callback
    must be handed the whole block; same iteration count and eigenvalues on every rank as on one rank."""
    n, lowest, nranks = 2999, 4, 3
    keep = []
    with fd.DavidsonEngine(n, lowest, None, gev=gev) as eng:
        set_ops(eng, user, gev, keep)
        lam1, _, it1 = eng.solve("DPR", 200, 1e-8, want_vectors=False)
    engs = [fd.DavidsonEngine(n, lowest, None, gev=gev, rank=r, nranks=nranks) for r in range(nranks)]
    handles = (C.c_void_p * nranks)(*[e.c.h for e in engs])
    assert fd.hip_lib().dav_local_group_join(handles, nranks) == 0
    out, err = [None] * nranks, [None] * nranks

    def work(r):
        try:
            set_ops(engs[r], user, gev, keep)
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
    for ctx in keep:
        user.user_op_destroy(ctx)
    assert all(x is None for x in err), err
    for lam, _, it in out:
        assert it == it1 and np.abs(lam - lam1).max() < 1e-10


def test_a_callback_that_fails_is_reported(user):
    bad = fd.engine_c.DEVICE_APPLY_FN(lambda *a: 7)
    with fd.CEngine(n=512, max_cols=32) as e:
        e.set_operator_device(0, bad, 0, np.arange(512.0) + 1)
        with pytest.raises(fd.DavidsonHipError, match="device operator returned 7"):
            e.init_basis(4)
