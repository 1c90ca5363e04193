"""Python mirror of the reference's Fortran interface, forwarding to the Fortran host library.

    generalized_eigensolver(matrix | callable, lowest, method, max_iterations, tolerance,
                            max_dim_sub=None, second_matrix | callable=None)
        -> (eigenvalues, eigenvectors, iters)

mirrors `call generalized_eigensolver(mtx, eigenvalues, eigenvectors, lowest, method, max_iterations,
tolerance, iters [, max_dim_sub] [, second_matrix])` of module davidson (reference:
src/davidson.f90:51-52, :277-278), outputs returned instead of passed.  Every call goes
Python -> libfortran_davidson_amd.so (Fortran driver loop) -> libdavidson_hip.so (HIP kernels).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import fortran_lib
from .engine_c import CEngine

_METHOD = {"DPR": 0, "GJD": 1}
_CB = C.CFUNCTYPE(None, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def generalized_eigensolver(matrix, lowest, method, max_iterations, tolerance, max_dim_sub=None,
                            second_matrix=None):
    lib = fortran_lib()
    iters = C.c_int(-1)
    evals = np.zeros(lowest)
    if callable(matrix):
        # matrix-free specific: first argument is the block-apply callback; `second_matrix` is the
        # (mandatory, src/davidson.f90:366,379) B callback; the eigenvector array fixes n.
        raise TypeError("matrix-free call needs n: use generalized_eigensolver_free(fun_A, n, ...)")
    a = _f(matrix)
    n = a.shape[0]
    evecs = np.zeros((n, lowest), order="F")
    b = _f(second_matrix) if second_matrix is not None else np.zeros((1, 1), order="F")
    lib.fd_dense_solve(C.c_int(n), _dp(a), C.c_int(0 if second_matrix is None else 1), _dp(b), C.c_int(lowest),
                       C.c_int(_METHOD.get(method, 2)), C.c_int(max_iterations), C.c_double(tolerance),
                       C.c_int(-1 if max_dim_sub is None else max_dim_sub), _dp(evals), _dp(evecs), C.byref(iters))
    return evals, evecs, iters.value


def generalized_eigensolver_free(fun_matrix_gemv, n, lowest, method, max_iterations, tolerance, max_dim_sub,
                                 fun_second_matrix_gemv):
    """Matrix-free specific with numpy callbacks X(n,k) -> Y(n,k) (reference: src/davidson.f90:277-337)."""
    lib = fortran_lib()

    def wrap(fn):
        def cb(nn, k, xp, yp):
            x = np.ctypeslib.as_array(xp, shape=(k, nn)).T
            y = np.ctypeslib.as_array(yp, shape=(k, nn))
            y[:, :] = np.asarray(fn(np.array(x, order="F"))).T
        return _CB(cb)

    fa, fb = wrap(fun_matrix_gemv), wrap(fun_second_matrix_gemv)
    evals = np.zeros(lowest)
    evecs = np.zeros((n, lowest), order="F")
    iters = C.c_int(-1)
    lib.fd_free_solve(C.c_int(n), fa, fb, C.c_int(lowest), C.c_int(max_iterations), C.c_double(tolerance),
                      C.c_int(10 * lowest if max_dim_sub is None else max_dim_sub), _dp(evals), _dp(evecs),
                      C.byref(iters))
    return evals, evecs, iters.value


class DavidsonEngine:
    """Device-resident problem (Fortran type `davidson_engine`): operators stay in HBM across solves.

    The third specific of the generic: `call generalized_eigensolver(engine, eigenvalues, eigenvectors,
    lowest, method, max_iterations, tolerance, iters, max_dim_sub)`.
    """

    def __init__(self, n, lowest, max_dim_sub=None, gev=False, device=0, rank=0, nranks=1, storage="full"):
        self.lib = fortran_lib()
        self.n, self.lowest = n, lowest
        self.max_dim = 10 * lowest if max_dim_sub is None else max_dim_sub
        self.gev = gev
        self.p = C.c_void_p(self.lib.fd_engine_create(C.c_int(n), C.c_int(lowest), C.c_int(self.max_dim),
                                                      C.c_int(1 if gev else 0), C.c_int(device), C.c_int(rank),
                                                      C.c_int(nranks)))
        self.c = CEngine(handle=self.lib.fd_engine_handle(self.p))
        if storage != "full":
            self.lib.fd_engine_set_storage(self.p, C.c_int({"full": 0, "symmetric": 1}[storage]))

    def close(self):
        if self.p:
            self.lib.fd_engine_destroy(self.p)
            self.p = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def comm_init(self, unique_id: bytes):
        self.lib.fd_engine_comm_init(self.p, C.create_string_buffer(unique_id, 128))

    def set_dense(self, which, matrix):
        a = _f(matrix)
        assert a.shape == (self.n, self.n)
        self.lib.fd_engine_set_dense(self.p, C.c_int(which), _dp(a))

    def set_correction_policy(self, policy):
        """"all" = the reference's policy (default); "unconverged" = opt-in: correct only the wanted pairs
        that have not converged; "locking" = opt-in (standard problems): converged wanted pairs are locked and the
        search space is kept orthogonal to them (Fortran: engine_set_correction_policy)."""
        self.lib.fd_engine_set_policy(self.p, C.c_int({"all": 0, "unconverged": 1, "locking": 2}[policy]))

    def set_inner_precision(self, bits):
        """32: the sweeps inside the GJD correction read an fp32 copy of the stored symmetric tiles (Fortran:
        engine_set_inner_precision); 64 (default): the reference's precision throughout."""
        self.lib.fd_engine_set_inner_precision(self.p, C.c_int(bits))

    def set_device_rr(self, on=True):
        """Rayleigh-Ritz on the device (Fortran: engine_set_device_rr); default off = host LAPACK as the reference."""
        self.lib.fd_engine_set_device_rr(self.p, C.c_int(1 if on else 0))

    def read_matrix(self, which, path, fmt="text"):
        """Operator from a file, streamed to HBM (Fortran: engine_read_matrix): "text" = the reference's
        write_matrix/read_matrix dump format, "f64" = raw row-major float64."""
        self.c.set_dense_file(which - 1, path, fmt)

    def _set_op(self, which, kind, seed, sparsity, diag_val):
        self.lib.fd_engine_set_operator(self.p, C.c_int(which), C.c_int(kind), C.c_int(seed), C.c_double(sparsity),
                                        C.c_int(0 if diag_val is None else 1),
                                        C.c_double(0.0 if diag_val is None else diag_val))

    def generate_diagonal_dominant(self, which, sparsity, diag_val=None, seed=1):
        """generate_diagonal_dominant(n, sparsity[, diag_val]) built directly in HBM."""
        self._set_op(which, 0, seed, sparsity, diag_val)

    def set_hashed_operator(self, which, sparsity, diag_val=None, seed=1):
        self._set_op(which, 1, seed, sparsity, diag_val)

    def set_harness_operator(self, which):
        self._set_op(which, 2, 0, 0.0, None)

    def set_identity(self, which):
        self._set_op(which, 3, 0, 0.0, None)

    def set_device_operator(self, which, fn, ctx, diag):
        """The caller's own operator as a block apply on device memory (engine_set_device_operator; which = 1 / 2)."""
        self.c.set_operator_device(which - 1, fn, ctx, diag)

    def solve(self, method="DPR", max_iterations=1000, tolerance=1e-8, want_vectors=True):
        evals = np.zeros(self.lowest)
        evecs = np.zeros((self.n, self.lowest) if want_vectors else (1, 1), order="F")
        iters = C.c_int(-1)
        self.lib.fd_engine_solve(self.p, C.c_int(self.lowest), C.c_int(_METHOD.get(method, 2)), C.c_int(max_iterations),
                                 C.c_double(tolerance), C.c_int(self.max_dim), _dp(evals),
                                 C.c_int(1 if want_vectors else 0), _dp(evecs), C.byref(iters))
        return evals, (evecs if want_vectors else None), iters.value


# ---- helper modules (array_utils / lapack_wrapper) --------------------------------------------------
def generate_diagonal_dominant(m, sparsity, diag_val=None, seed=1):
    out = np.zeros((m, m), order="F")
    fortran_lib().fd_generate_diagonal_dominant(C.c_int(m), C.c_double(sparsity), C.c_int(0 if diag_val is None else 1),
                                                C.c_double(0.0 if diag_val is None else diag_val), C.c_int(seed), _dp(out))
    return out


def lapack_generalized_eigensolver(mtx, stx=None):
    mtx = _f(mtx)
    n = mtx.shape[0]
    s = _f(stx) if stx is not None else np.zeros((1, 1), order="F")
    w = np.zeros(n)
    v = np.zeros((n, n), order="F")
    fortran_lib().fd_lapack_eigensolver(C.c_int(n), _dp(mtx), C.c_int(0 if stx is None else 1), _dp(s), _dp(w), _dp(v))
    return w, v


def lapack_rayleigh_ritz(mtx, nvec, stx=None):
    """The Rayleigh-Ritz solver of the outer loop: lowest `nvec` pairs (first nvec entries / columns valid)."""
    mtx = _f(mtx)
    n = mtx.shape[0]
    s = _f(stx) if stx is not None else np.zeros((1, 1), order="F")
    w = np.zeros(n)
    v = np.zeros((n, n), order="F")
    fortran_lib().fd_lapack_rayleigh_ritz(C.c_int(n), _dp(mtx), C.c_int(0 if stx is None else 1), _dp(s), C.c_int(nvec),
                                          _dp(w), _dp(v))
    return w[:nvec], v[:, :nvec]


def lapack_qr(basis):
    q = _f(basis).copy(order="F")
    fortran_lib().fd_lapack_qr(C.c_int(q.shape[0]), C.c_int(q.shape[1]), _dp(q))
    return q


def lapack_solver(arr, brr):
    a = _f(arr).copy(order="F")
    b = np.array(brr, dtype=np.float64).reshape(-1, 1).copy(order="F")
    fortran_lib().fd_lapack_solver(C.c_int(a.shape[0]), _dp(a), _dp(b))
    return b[:, 0]


def lapack_matmul(transA, transB, arr, brr):
    a, b = _f(arr), _f(brr)
    m = a.shape[1] if transA == "T" else a.shape[0]
    k = a.shape[0] if transA == "T" else a.shape[1]
    n = b.shape[0] if transB == "T" else b.shape[1]
    c = np.zeros((m, n), order="F")
    fortran_lib().fd_lapack_matmul(C.c_int(transA == "T"), C.c_int(transB == "T"), C.c_int(m), C.c_int(k), C.c_int(n),
                                   _dp(a), _dp(b), _dp(c))
    return c


def lapack_sort(id_, vector):
    v = np.array(vector, dtype=np.float64)
    keys = np.zeros(v.size, dtype=np.int32)
    fortran_lib().fd_lapack_sort(C.c_int(v.size), C.c_int(id_ == "D"), _dp(v), keys.ctypes.data_as(C.POINTER(C.c_int)))
    return keys, v


def generate_preconditioner(diag, dim_sub):
    d = np.array(diag, dtype=np.float64)
    out = np.zeros((d.size, dim_sub), order="F")
    fortran_lib().fd_generate_preconditioner(C.c_int(d.size), _dp(d), C.c_int(dim_sub), _dp(out))
    return out


def norm(v):
    v = np.ascontiguousarray(v, dtype=np.float64)
    res = C.c_double()
    fortran_lib().fd_norm(C.c_int(v.size), _dp(v), C.byref(res))
    return res.value
