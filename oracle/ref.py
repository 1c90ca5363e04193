"""TEST INFRASTRUCTURE ONLY - ctypes access to oracle/_ref/libref_davidson.so.

That library is the reference itself (NLESC-JCER/Fortran_Davidson) compiled by oracle/build_ref.sh
plus our bind(C) driver (oracle/ref_driver.f90).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.  The product never does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_davidson.so")
_lib = None

_CB = C.CFUNCTYPE(None, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double))


def available() -> bool:
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError(f"{_PATH} missing: run oracle/build_ref.sh where /root/reference exists")
        _lib = C.CDLL(_PATH, mode=C.RTLD_LOCAL)
    return _lib


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def dense_solve(a, lowest, method="DPR", max_it=1000, tol=1e-8, max_dim=None, b=None):
    """The reference's generalized_eigensolver (dense), davidson.f90:51-246."""
    a = _f(a)
    n = a.shape[0]
    bb = _f(b) if b is not None else np.zeros((1, 1), order="F")
    evals = np.zeros(lowest)
    evecs = np.zeros((n, lowest), order="F")
    iters = C.c_int(-1)
    lib().ref_dense_solve(
        C.c_int(n), _p(a), C.c_int(0 if b is None else 1), _p(bb), C.c_int(lowest),
        C.c_int({"DPR": 0, "GJD": 1}[method]), C.c_int(max_it), C.c_double(tol),
        C.c_int(-1 if max_dim is None else max_dim), _p(evals), _p(evecs), C.byref(iters))
    return evals, evecs, iters.value


def generate(n, sparsity, seed=1, diag_val=None):
    """generate_diagonal_dominant of the oracle (davidson_oracle.py, bit-identical) produced under OpenMP into a fresh array:
    parallel first touch - the input of the CPU baseline (ref_driver.f90: ref_generate)."""
    a = np.empty((n, n), order="F")
    lib().ref_generate(C.c_int(n), C.c_double(sparsity), C.c_int(seed), C.c_int(0 if diag_val is None else 1),
                       C.c_double(0.0 if diag_val is None else diag_val), _p(a))
    return a


def stream_sum(a):
    """sum of all entries under OpenMP (ref_driver.f90: ref_stream_sum) - a parallel read of the matrix, for its GB/s"""
    out = C.c_double(0.0)
    lib().ref_stream_sum(C.c_int(a.shape[0]), _p(a), C.byref(out))
    return out.value


def gemv_omp(a, x):
    """y = A x under OpenMP (ref_driver.f90: ref_gemv_omp): the yardstick next to MKL's DGEMV in the CPU baseline"""
    y = np.zeros(a.shape[0])
    lib().ref_gemv_omp(C.c_int(a.shape[0]), _p(a), _p(np.ascontiguousarray(x, dtype=np.float64)), _p(y))
    return y


def free_solve_harness(n, lowest, max_it=1000, tol=1e-8, max_dim=20):
    """Reference matrix-free solve with its own test operators (tests/test_utils.f90:11-116)."""
    evals = np.zeros(lowest)
    evecs = np.zeros((n, lowest), order="F")
    iters = C.c_int(-1)
    lib().ref_free_solve_harness(C.c_int(n), C.c_int(lowest), C.c_int(max_it), C.c_double(tol),
                                 C.c_int(max_dim), _p(evals), _p(evecs), C.byref(iters))
    return evals, evecs, iters.value


def free_solve_benchmark(n=1000, lowest=3, max_it=1000, tol=1e-8, max_dim=20):
    """The reference's benchmark program as a call (benchmark_free.f90:80-111): A = the cos row generator, B = I, DPR; defaults =
    that program's parameters."""
    evals = np.zeros(lowest)
    evecs = np.zeros((n, lowest), order="F")
    iters = C.c_int(-1)
    lib().ref_free_solve_benchmark(C.c_int(n), C.c_int(lowest), C.c_int(max_it), C.c_double(tol),
                                   C.c_int(max_dim), _p(evals), _p(evecs), C.byref(iters))
    return evals, evecs, iters.value


def free_solve_callbacks(n, apply_a, apply_b, lowest, max_it=1000, tol=1e-8, max_dim=20):
    """Reference matrix-free solve (davidson.f90:277-460) with numpy callbacks X(n,k)->Y(n,k)."""
    def wrap(fn):
        def cb(nn, k, xp, yp):
            x = np.ctypeslib.as_array(xp, shape=(k, nn)).T  # column-major (n,k)
            y = np.ctypeslib.as_array(yp, shape=(k, nn))
            y[:, :] = np.asarray(fn(np.array(x))).T
        return _CB(cb)
    fa, fb = wrap(apply_a), wrap(apply_b)
    evals = np.zeros(lowest)
    evecs = np.zeros((n, lowest), order="F")
    iters = C.c_int(-1)
    lib().ref_free_solve_cb(C.c_int(n), fa, fb, C.c_int(lowest), C.c_int(max_it), C.c_double(tol),
                            C.c_int(max_dim), _p(evals), _p(evecs), C.byref(iters))
    return evals, evecs, iters.value


def harness_matrices(n):
    mtx = np.zeros((n, n), order="F")
    stx = np.zeros((n, n), order="F")
    lib().ref_harness_matrices(C.c_int(n), _p(mtx), _p(stx))
    return mtx, stx


def lapack_qr(basis):
    q = _f(basis).copy(order="F")
    lib().ref_lapack_qr(C.c_int(q.shape[0]), C.c_int(q.shape[1]), _p(q))
    return q


def lapack_eigensolver(mtx, stx=None):
    mtx = _f(mtx)
    n = mtx.shape[0]
    s = _f(stx) if stx is not None else np.zeros((1, 1), order="F")
    evals = np.zeros(n)
    evecs = np.zeros((n, n), order="F")
    lib().ref_lapack_eigensolver(C.c_int(n), _p(mtx), C.c_int(0 if stx is None else 1), _p(s),
                                 _p(evals), _p(evecs))
    return evals, evecs


def generate_preconditioner(diag, dim_sub):
    d = np.ascontiguousarray(diag, dtype=np.float64)
    out = np.zeros((d.size, dim_sub), order="F")
    lib().ref_generate_preconditioner(C.c_int(d.size), _p(d), C.c_int(dim_sub), _p(out))
    return out


def lapack_solver(arr, brr):
    a = _f(arr).copy(order="F")
    b = np.array(brr, dtype=np.float64).reshape(-1, 1).copy(order="F")
    lib().ref_lapack_solver(C.c_int(a.shape[0]), _p(a), _p(b))
    return b[:, 0]
