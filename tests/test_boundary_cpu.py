"""CPU: the drop-in boundary loads and exports what include/davidson_hip.h declares; the Fortran host
helper modules (lapack_wrapper / array_utils mirrors) agree with the oracle.  No compute on a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import fortran_davidson_amd as fd
from oracle import davidson_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "davidson_hip.h")).read()
    names = sorted(set(re.findall(r"\b(dav_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 30
    lib = fd.hip_lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.dav_version() >= 100


def test_fortran_host_exports_api_doors():
    lib = fd.fortran_lib()
    for name in ["fd_dense_solve", "fd_free_solve", "fd_engine_create", "fd_engine_solve", "fd_engine_destroy",
                 "fd_engine_set_operator", "fd_engine_set_dense", "fd_lapack_qr", "fd_lapack_eigensolver"]:
        assert hasattr(lib, name), name
    # the Fortran module procedures themselves are in the library (drop-in link target)
    out = os.popen(f"nm -D {os.path.join(ROOT, 'fortran_davidson_amd', 'lib', 'libfortran_davidson_amd.so')}").read()
    for mod_proc in ["davidson_dense", "davidson_free", "generalized_eigensolver_dense",
                     "generalized_eigensolver_free", "free_matmul", "lapack_qr", "generate_preconditioner"]:
        assert mod_proc in out.lower(), mod_proc


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(fd.DavidsonHipError):
        fd.CEngine(n=64, max_cols=16)


def test_host_generator_matches_oracle_bitwise():
    for n, sp, dv, seed in [(64, 1e-3, None, 1), (50, 1e-3, 1.0, 2), (33, 5e-2, None, 9)]:
        assert np.array_equal(fd.generate_diagonal_dominant(n, sp, dv, seed),
                              O.generate_diagonal_dominant(n, sp, dv, seed))


def test_lapack_wrapper_mirrors(golden):
    _, arrays = golden
    A = O.generate_diagonal_dominant(50, 1e-3, seed=7)
    B = O.generate_diagonal_dominant(50, 1e-3, 1.0, seed=8)
    w, v = fd.lapack_generalized_eigensolver(A)
    assert np.allclose(w, arrays["lapack__dsyev_w"], atol=1e-12)
    assert np.allclose(np.abs(v), np.abs(arrays["lapack__dsyev_v"]), atol=1e-8)
    w, v = fd.lapack_generalized_eigensolver(A, B)
    assert np.allclose(w, arrays["lapack__dsygv_w"], atol=1e-12)
    q = fd.lapack_qr(A[:, :20])
    assert np.allclose(q, arrays["lapack__qr_q"], atol=1e-12)
    x = np.random.default_rng(0).standard_normal(50)
    assert np.allclose(fd.lapack_solver(A, A @ x), x, atol=1e-10)
    assert np.allclose(fd.lapack_matmul("T", "N", A[:, :7], B[:, :5]), A[:, :7].T @ B[:, :5], atol=1e-13)
    assert np.allclose(fd.lapack_matmul("N", "T", A[:, :7], B[:, :7]), A[:, :7] @ B[:, :7].T, atol=1e-13)


def test_sort_and_preconditioner():
    rng = np.random.default_rng(3)
    d = rng.standard_normal(200)
    keys, s = fd.lapack_sort("I", d)
    assert np.array_equal(s, np.sort(d))
    assert np.array_equal(s[keys - 1], d)                     # keys(i) = rank of the original entry i
    keys, s = fd.lapack_sort("D", d)
    assert np.array_equal(s, np.sort(d)[::-1])
    # duplicates: stable
    keys, s = fd.lapack_sort("I", np.array([2.0, 1.0, 2.0, 1.0]))
    assert list(keys) == [3, 1, 4, 2]
    pre = fd.generate_preconditioner(d, 6)
    assert np.array_equal(pre, O.generate_preconditioner(d, 6))
    assert fd.norm(np.array([3.0, 4.0])) == 5.0


@pytest.mark.parametrize("n,nvec", [(40, 40), (40, 6), (48, 48), (64, 8), (150, 150), (150, 20), (300, 64)])
@pytest.mark.parametrize("gev", [False, True])
def test_rayleigh_ritz_solver_all_routes_against_scipy(n, nvec, gev):
    """lapack_rayleigh_ritz: DSYEV/DSYGV below order 48 (the reference's route), divide and conquer above,
    MRRR on a leading subset (Cholesky-reduced when generalized) - same eigenpairs, DSYGV normalisation."""
    import scipy.linalg
    from fortran_davidson_amd.solver import lapack_rayleigh_ritz
    rng = np.random.default_rng(n + nvec)
    H = rng.standard_normal((n, n)); H = (H + H.T) / 2 + np.diag(np.arange(n))
    S = None
    if gev:
        S = rng.standard_normal((n, n)) * 0.02; S = np.eye(n) + (S + S.T) / 2
    w, Y = lapack_rayleigh_ritz(H, nvec, S)
    ref = scipy.linalg.eigh(H, S, eigvals_only=True)[:nvec]
    assert np.abs(w - ref).max() < 1e-10
    SY = Y if S is None else S @ Y
    assert np.abs(H @ Y - SY * w[None, :]).max() < 1e-9
    assert np.abs(Y.T @ SY - np.eye(nvec)).max() < 1e-10


def test_product_library_is_built_without_the_test_transports():
    """lib/libdavidson_hip.so (what a user links) carries no loopback / shared-memory transport: its two entry points are
    stubs that fail; lib/test/libdavidson_hip.so (same sources, -DDAV_TEST_TRANSPORTS=1, same soname) is what pytest loads."""
    import ctypes as C
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prod = os.path.join(root, "fortran_davidson_amd", "lib", "libdavidson_hip.so")
    test = os.path.join(root, "fortran_davidson_amd", "lib", "test", "libdavidson_hip.so")
    for path in (prod, test):
        assert os.path.exists(path), path
        dyn = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
        assert "soname: [libdavidson_hip.so]" in dyn
    stub_msg = b"built without DAV_TEST_TRANSPORTS"
    assert stub_msg in open(prod, "rb").read()
    assert stub_msg not in open(test, "rb").read()
    assert os.environ.get("DAVIDSON_HIP_LIB") == test          # conftest.py
    # the stubs fail without touching a GPU
    lib = C.CDLL(prod, mode=C.RTLD_LOCAL)
    lib.dav_last_error.restype = C.c_char_p
    assert lib.dav_local_group_join(None, 0) != 0 and stub_msg in lib.dav_last_error()
