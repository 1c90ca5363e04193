"""TEST INFRASTRUCTURE ONLY - numpy/scipy restatement of the reference's Davidson hot path.

This file is the ORACLE for the MI355X engine.  It restates, operation by operation, what
NLESC-JCER/Fortran_Davidson computes on its CPU+LAPACK path; every function cites the reference
file:line it follows (paths relative to /root/reference).  It is never imported by the product
(`fortran_davidson_amd/`); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
use it, and only as the checker.

Parity pinning (see tests/test_oracle_golden.py and oracle/gen_golden.py):
  * against the reference itself compiled here (oracle/_ref, built by oracle/build_ref.sh) on the
    reference's own data file src/tests/matrix.txt and on matrices of our counter-based generator -
    eigenvalues, residual norms, iteration counts and basis-width trajectories;
  * against the known answers SURVEY.md 8(c) records for matrix.txt and the matrix-free harness.

The only deliberate deviations from the reference (both documented in SURVEY.md Appendix B):
  * `generate_diagonal_dominant` uses OUR counter-based generator instead of the compiler-specific,
    unseeded `random_number` stream (array_utils.f90:96) so that host and device build bit-identical
    inputs.  Semantics (symmetric, off-diagonal U[0,1)*sparsity, diagonal i or diag_val) are those
    of array_utils.f90:86-113.
  * `lapack_sort` + `search_key` are a stable argsort (identical for distinct diagonals; the
    reference's O(n^2) key search is undefined for duplicates, lapack_wrapper.f90:384-390).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import lapack as _la

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


# --------------------------------------------------------------------------------------------
# input generator (semantics: array_utils.f90:86-113; random stream: ours, counter based)
# --------------------------------------------------------------------------------------------
def _splitmix64(z):
    """splitmix64 finaliser on uint64 values (arithmetic wraps modulo 2^64)"""
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, lo, hi):
    """u(seed, lo, hi) in [0,1): lo<=hi are 0-based indices.  53-bit mantissa, exact in fp64."""
    lo = np.asarray(lo, dtype=np.uint64)
    hi = np.asarray(hi, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = (lo << np.uint64(32)) + hi + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
    z = _splitmix64(key)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def generate_diagonal_dominant(m, sparsity, diag_val=None, seed=1, rows=None):
    """array_utils.f90:86-113: arr = U*sparsity, symmetrised from the upper triangle
    (arr(i,j) = arr(j,i) for i>j), diagonal = i (1-based) or diag_val.
    `rows=(r0,r1)` returns only that row slab (used by the multi-GPU tests); `rows=<integer array>` returns
    those rows (0-based, any order) - how the full-size tests obtain rows of a matrix that does not fit the host."""
    if rows is None:
        ii = np.arange(0, m)
    elif isinstance(rows, tuple):
        ii = np.arange(rows[0], rows[1])
    else:
        ii = np.asarray(rows, dtype=np.int64).ravel()
    i = ii.astype(np.uint64)[:, None]
    a = np.empty((len(ii), m), order="F")
    step = max(256, (1 << 21) // max(len(ii), 1))        # column chunks: the temporaries stay small
    for c0 in range(0, m, step):
        j = np.arange(c0, min(m, c0 + step), dtype=np.uint64)[None, :]
        a[:, c0:c0 + j.shape[1]] = uniform01(seed, np.minimum(i, j), np.maximum(i, j)) * float(sparsity)
    a[np.arange(len(ii)), ii] = (ii + 1).astype(np.float64) if diag_val is None else float(diag_val)
    return a


# --------------------------------------------------------------------------------------------
# array_utils.f90 helpers
# --------------------------------------------------------------------------------------------
def norm(v):
    """array_utils.f90:46-53: sqrt(sum(v**2)), no scaling."""
    v = np.asarray(v, dtype=np.float64)
    return float(np.sqrt(np.sum(v * v)))


def diagonal(matrix):
    """array_utils.f90:115-134."""
    return np.array(np.diag(matrix), dtype=np.float64)


def lowest_diagonal_indices(diag, dim_sub):
    """lapack_wrapper.f90:367-392 (DLASRT 'I' + key recovery) followed by
    array_utils.f90:162-179 (search_key): index of the i-th smallest diagonal entry, i=1..dim_sub.
    Stable argsort: identical to the reference whenever the entries are distinct."""
    return np.argsort(np.asarray(diag, dtype=np.float64), kind="stable")[:dim_sub]


def generate_preconditioner(diag, dim_sub):
    """array_utils.f90:136-160: V0(:, i) = e_{index of i-th smallest diagonal entry}."""
    n = len(diag)
    v = np.zeros((n, dim_sub), order="F")
    idx = lowest_diagonal_indices(diag, dim_sub)
    v[idx, np.arange(dim_sub)] = 1.0
    return v


def concatenate(arr, brr):
    """array_utils.f90:55-84."""
    return np.asfortranarray(np.hstack([arr, brr]))


# --------------------------------------------------------------------------------------------
# lapack_wrapper.f90
# --------------------------------------------------------------------------------------------
def lapack_generalized_eigensolver(mtx, stx=None):
    """lapack_wrapper.f90:14-91: DSYEV('V','U') or DSYGV(itype=1,'V','U'); all eigenpairs, ascending."""
    a = np.array(mtx, dtype=np.float64, order="F")
    if stx is None:
        w, v, info = _la.dsyev(a, compute_v=1, lower=0)
        _check(info, "DSYEV")
    else:
        b = np.array(stx, dtype=np.float64, order="F")
        w, v, info = _la.dsygv(a, b, itype=1, jobz="V", uplo="U")
        _check(info, "DSYGV")
    return w, v


def lapack_qr(basis):
    """lapack_wrapper.f90:176-236: DGEQRF + DORGQR, explicit thin Q of the whole basis."""
    a = np.array(basis, dtype=np.float64, order="F")
    qr, tau, _, info = _la.dgeqrf(a)
    _check(info, "DGEQRF")
    q, _, info = _la.dorgqr(qr, tau)
    _check(info, "DORGQR")
    return q


def lapack_solver(arr, brr):
    """lapack_wrapper.f90:238-277: DSYSV('U'), on info>0 patch the zero pivot with tiny() and retry.
    As in the reference the retry re-runs DSYSV on the *already factorised* array (arr is
    overwritten in place by the first call, lapack_wrapper.f90:266-271)."""
    a = np.array(arr, dtype=np.float64, order="F")
    b = np.array(brr, dtype=np.float64).reshape(-1, 1).copy(order="F")
    udut, ipiv, x, info = _la.dsysv(a, b, lower=0)
    if info > 0:
        udut = np.array(udut, order="F")
        udut[info - 1, info - 1] = np.finfo(np.float64).tiny
        udut, ipiv, x, info = _la.dsysv(udut, b, lower=0)
        _check(info, "DSYSV")
    elif info < 0:
        _check(info, "DSYSV")
    return x[:, 0]


def _check(info, name):
    """lapack_wrapper.f90:395-408 (error stop -> exception)."""
    if info != 0:
        raise RuntimeError(f"call to subroutine: {name} has failed! info: {info}")


# --------------------------------------------------------------------------------------------
# correction methods, davidson.f90:630-752
# --------------------------------------------------------------------------------------------
def compute_DPR_generalized_dense(matrix, eigenvalues, residues, second_matrix=None):
    """davidson.f90:673-698: t_ij = r_ij / (theta_j - A_ii)  |  / (theta_j * B_ii - A_ii)."""
    da = np.diag(matrix)[:, None]
    th = np.asarray(eigenvalues)[None, : residues.shape[1]]
    if second_matrix is not None:
        db = np.diag(second_matrix)[:, None]
        return residues / (th * db - da)
    return residues / (th - da)


def compute_GJD_generalized_dense(matrix, eigenvalues, ritz_vectors, residues, second_matrix=None):
    """davidson.f90:700-734: per Ritz pair k solve P (A - theta_k B) P t = -r_k with
    P = I - x_k x_k^T (Euclidean projector, davidson.f90:721) through lapack_solver."""
    n = matrix.shape[0]
    out = np.zeros((n, ritz_vectors.shape[1]), order="F")
    for k in range(ritz_vectors.shape[1]):
        rs = ritz_vectors[:, k:k + 1]
        xs = np.eye(n) - rs @ rs.T
        if second_matrix is not None:
            ys = matrix - eigenvalues[k] * second_matrix
        else:
            ys = matrix - eigenvalues[k] * np.eye(n)            # davidson.f90:736-750
        arr = xs @ (ys @ xs)
        out[:, k] = lapack_solver(arr, -residues[:, k])
    return out


def compute_DPR_free(eigenvalues, residues, diag_matrix, diag_second_matrix):
    """davidson.f90:463-488."""
    th = np.asarray(eigenvalues)[None, : residues.shape[1]]
    return residues / (th * np.asarray(diag_second_matrix)[:, None] - np.asarray(diag_matrix)[:, None])


# --------------------------------------------------------------------------------------------
# dense solver, davidson.f90:51-246
# --------------------------------------------------------------------------------------------
class Trace:
    """What the parity tests compare besides eigenvalues: per-iteration basis width, the residual
    norms of the first `lowest` pairs, and the exit iteration."""

    def __init__(self):
        self.widths = []
        self.errors = []
        self.converged = False


def generalized_eigensolver_dense(matrix, lowest, method, max_iterations, tolerance,
                                  max_dim_sub=None, second_matrix=None, trace=None):
    """davidson.f90:51-246.  Returns (eigenvalues, eigenvectors, iters)."""
    A = np.asarray(matrix, dtype=np.float64)
    B = None if second_matrix is None else np.asarray(second_matrix, dtype=np.float64)
    n = A.shape[0]
    initial_dimension = 2 * lowest                                   # :108
    max_dim = max_dim_sub if max_dim_sub is not None else 10 * lowest  # :115-119
    gev = B is not None                                              # :122
    has_converged = np.zeros(lowest, dtype=bool)                     # :112
    if method not in ("DPR", "GJD"):
        raise ValueError("method must be DPR or GJD")                # :656-669 (no default case)

    d = diagonal(A)                                                  # :127
    V = generate_preconditioner(d, initial_dimension)                # :128
    H = V.T @ (A @ V)                                                # :131
    S = V.T @ (B @ V) if gev else None                               # :133-135

    eigenvalues = np.zeros(lowest)
    eigenvectors = np.zeros((n, lowest), order="F")
    iters = max_iterations + 1                                       # :232-235
    for i in range(1, max_iterations + 1):                           # :138
        theta, Y = lapack_generalized_eigensolver(H, S)              # :152-156
        X = V @ Y                                                    # :159
        m = V.shape[1]
        R = np.empty((n, m), order="F")
        for j in range(m):                                           # :163-170
            guess = theta[j] * (B @ X[:, j]) if gev else theta[j] * X[:, j]
            R[:, j] = A @ X[:, j] - guess
        errors = np.array([norm(R[:, j]) for j in range(lowest)])    # :173-178
        has_converged |= errors < tolerance                          # sticky, :176
        eigenvalues = theta[:lowest].copy()                          # :186
        eigenvectors = np.asfortranarray(X[:, :lowest])              # :187
        if trace is not None:
            trace.widths.append(m)
            trace.errors.append(errors)
        if has_converged.all():                                      # :189-192
            iters = i
            if trace is not None:
                trace.converged = True
            break
        if m <= max_dim:                                             # :195
            if method == "DPR":
                T = compute_DPR_generalized_dense(A, theta, R, B)
            else:
                T = compute_GJD_generalized_dense(A, theta, X, R, B)
            V = lapack_qr(concatenate(V, T))                         # :210-213
        else:
            V = V @ Y[:, :initial_dimension]                         # :218
        H = V.T @ (A @ V)                                            # :223
        if gev:
            S = V.T @ (B @ V)                                        # :226
    return eigenvalues, eigenvectors, iters


def generalized_eigensolver_dense_unconverged(matrix, lowest, method, max_iterations, tolerance,
                                              max_dim_sub=None, second_matrix=None, trace=None):
    """NOT in the reference: CPU statement of the engine's OPT-IN correction policy "unconverged"
    (fortran_davidson_amd/fortran/davidson.f90, POLICY_UNCONVERGED), kept here as its checker.
    Same building blocks as generalized_eigensolver_dense above (davidson.f90:51-246) with three changes:
    residues and corrections only for the `lowest` wanted pairs, and among them only for those whose residual
    is still >= tolerance; convergence tested on all wanted pairs at once; the basis grows while
    m + lowest <= max_dim (a basis of 2*lowest columns always grows), else collapses to 2*lowest (:218)."""
    A = np.asarray(matrix, dtype=np.float64)
    B = None if second_matrix is None else np.asarray(second_matrix, dtype=np.float64)
    n = A.shape[0]
    initial_dimension = 2 * lowest
    max_dim = max_dim_sub if max_dim_sub is not None else 10 * lowest
    gev = B is not None
    if method not in ("DPR", "GJD"):
        raise ValueError("method must be DPR or GJD")
    V = generate_preconditioner(diagonal(A), initial_dimension)
    H = V.T @ (A @ V)
    S = V.T @ (B @ V) if gev else None
    eigenvalues = np.zeros(lowest)
    eigenvectors = np.zeros((n, lowest), order="F")
    iters = max_iterations + 1
    for i in range(1, max_iterations + 1):
        theta, Y = lapack_generalized_eigensolver(H, S)
        m = V.shape[1]
        X = np.asfortranarray(V @ Y[:, :lowest])
        BX = B @ X if gev else X
        R = np.asfortranarray(A @ X - BX * theta[None, :lowest])
        errors = np.array([norm(R[:, j]) for j in range(lowest)])
        eigenvalues = theta[:lowest].copy()
        eigenvectors = X
        if trace is not None:
            trace.widths.append(m)
            trace.errors.append(errors)
        if (errors < tolerance).all():
            iters = i
            if trace is not None:
                trace.converged = True
            break
        if m + lowest <= max_dim or m <= initial_dimension:
            sel = np.nonzero(errors >= tolerance)[0]
            if method == "DPR":
                T = compute_DPR_generalized_dense(A, theta[sel], np.asfortranarray(R[:, sel]), B)
            else:
                T = compute_GJD_generalized_dense(A, theta[sel], np.asfortranarray(X[:, sel]),
                                                  np.asfortranarray(R[:, sel]), B)
            V = lapack_qr(concatenate(V, T))
        else:
            V = V @ Y[:, :initial_dimension]
        H = V.T @ (A @ V)
        if gev:
            S = V.T @ (B @ V)
    return eigenvalues, eigenvectors, iters


def generalized_eigensolver_dense_locking(matrix, lowest, method, max_iterations, tolerance, max_dim_sub=None, trace=None,
                                          second_matrix=None):
    """NOT in the reference: CPU statement of the engine's OPT-IN correction policy "locking" (the deflation the
    reference's header cites, src/davidson.f90:7-8, and never implements; fortran_davidson_amd/fortran/davidson.f90,
    POLICY_LOCKING), kept here as its checker.  Same building blocks as generalized_eigensolver_dense (davidson.f90:51-246):
      * a wanted Ritz pair whose residual is below the tolerance is LOCKED: its Ritz vector joins Q, its value is final, and the
        active basis is rotated to the remaining Ritz vectors (V <- V Y(:, not locked));
      * the search space is kept orthogonal to the GUARD vectors U = B Q (standard problems: U = Q): the eigenvectors still wanted are
        B-orthogonal to the locked ones, x^T B q = 0, i.e. they lie in the complement of span(B Q) - and for a locked pair
        V^T A q = lambda V^T B q = 0 there, so the projected pencil of the active basis is decoupled from what is locked;
      * the Rayleigh-Ritz problem is that of the active basis alone (H = V^T A V, generalized: S = V^T B V), for the
        lowest - len(Q) pairs still wanted;
      * corrections only for the wanted pairs that are not locked; the new block is orthonormalised against U AND V (QR of
        [U V T], davidson.f90:210-213);
      * the active basis grows while m + wanted <= max_dim (one of 2*wanted columns or fewer always grows), else it collapses to its
        2*wanted lowest Ritz vectors (:218); the solve ends when `lowest` pairs are locked; eigenvalues returned in ascending order."""
    A = np.asarray(matrix, dtype=np.float64)
    B = None if second_matrix is None else np.asarray(second_matrix, dtype=np.float64)
    gev = B is not None
    n = A.shape[0]
    max_dim = max_dim_sub if max_dim_sub is not None else 10 * lowest
    if method not in ("DPR", "GJD"):
        raise ValueError("method must be DPR or GJD")
    V = generate_preconditioner(diagonal(A), 2 * lowest)
    Q = np.zeros((n, 0), order="F")

    def guards():
        """orthonormal basis of span(B Q) (standard problems: Q itself, orthonormal already)"""
        if Q.shape[1] == 0:
            return Q
        return lapack_qr(np.asfortranarray(B @ Q)) if gev else Q

    def complement(block):
        """orthonormal basis of the columns of `block` made orthogonal to the guards (QR of [U block])"""
        U = guards()
        W = lapack_qr(concatenate(U, block) if U.shape[1] else np.asfortranarray(block))
        return np.asfortranarray(W[:, U.shape[1]:])

    locked = []
    iters = max_iterations + 1
    theta = np.zeros(0)
    X = np.zeros((n, 0))
    for i in range(1, max_iterations + 1):
        want = lowest - len(locked)
        H = V.T @ (A @ V)
        theta, Y = lapack_generalized_eigensolver(H, V.T @ (B @ V) if gev else None)
        m = V.shape[1]
        X = np.asfortranarray(V @ Y[:, :want])
        BX = np.asfortranarray(B @ X) if gev else X
        R = np.asfortranarray(A @ X - BX * theta[None, :want])
        errors = np.array([norm(R[:, j]) for j in range(want)])
        if trace is not None:
            trace.widths.append(m + len(locked))
            trace.errors.append(errors)
        conv = errors < tolerance
        rest = [j for j in range(m) if not (j < want and conv[j])]          # Ritz vectors that stay active
        if conv.any():
            Q = np.asfortranarray(np.hstack([Q, X[:, conv]]))
            locked.extend(float(t) for t in theta[:want][conv])
        if len(locked) == lowest:
            iters = i
            if trace is not None:
                trace.converged = True
            break
        want_new = lowest - len(locked)
        sel = np.nonzero(~conv)[0]
        m_rest = len(rest)
        if m_rest + want_new <= max_dim or m_rest <= 2 * want_new:
            if conv.any():
                V = np.asfortranarray(V @ Y[:, rest])
            if method == "DPR":
                T = compute_DPR_generalized_dense(A, theta[sel], np.asfortranarray(R[:, sel]), B)
            else:
                T = compute_GJD_generalized_dense(A, theta[sel], np.asfortranarray(X[:, sel]), np.asfortranarray(R[:, sel]), B)
            T = T[:, :max(0, n - len(locked) - V.shape[1])]                  # never more columns than the space has left
            V = complement(concatenate(V, T))
        else:
            V = np.asfortranarray(V @ Y[:, rest[:2 * want_new]])
            if gev:
                V = complement(V)                                            # S-orthonormal Ritz vectors: Euclidean-orthonormal again
    order = np.argsort(np.array(locked)) if len(locked) == lowest else None
    if order is not None:
        return np.array(locked)[order], np.asfortranarray(Q[:, order]), iters
    # not converged: what is locked plus the current Ritz pairs of the active basis
    lam = np.concatenate([np.array(locked), theta[:lowest - len(locked)]])
    vec = np.hstack([Q, X[:, :lowest - len(locked)]])
    order = np.argsort(lam)
    return lam[order], np.asfortranarray(vec[:, order]), iters


# --------------------------------------------------------------------------------------------
# matrix-free solver, davidson.f90:277-460
# --------------------------------------------------------------------------------------------
def extract_diagonal_free(fun, dim):
    """davidson.f90:490-523: N unit-vector applies."""
    out = np.zeros(dim)
    for ii in range(dim):
        e = np.zeros((dim, 1), order="F")
        e[ii, 0] = 1.0
        out[ii] = fun(e)[ii, 0]
    return out


def generalized_eigensolver_free(fun_matrix_gemv, dim_matrix, lowest, max_iterations, tolerance,
                                 max_dim_sub, fun_second_matrix_gemv, trace=None,
                                 diag_matrix=None, diag_second_matrix=None):
    """davidson.f90:277-460 (always generalized, always DPR, non-sticky convergence).
    Returns (eigenvalues, ritz_vectors, iters); iters = max_iterations+1 when not converged
    (the reference leaves it unset, davidson.f90:444-446 - SURVEY Appendix B)."""
    initial_dimension = 2 * lowest                                            # :352
    max_dim = max_dim_sub if max_dim_sub is not None else 10 * lowest         # :355-359
    if diag_matrix is None:
        diag_matrix = extract_diagonal_free(fun_matrix_gemv, dim_matrix)      # :365
    if diag_second_matrix is None:
        diag_second_matrix = extract_diagonal_free(fun_second_matrix_gemv, dim_matrix)  # :366
    V = generate_preconditioner(np.array(diag_matrix), initial_dimension)     # :371-372
    iters = max_iterations + 1
    theta = None
    ritz = None
    for i in range(1, max_iterations + 1):                                    # :375
        AV = fun_matrix_gemv(V)                                               # :378
        BV = fun_second_matrix_gemv(V)                                        # :379
        H = V.T @ AV                                                          # :380
        S = V.T @ BV                                                          # :381
        theta, Y = lapack_generalized_eigensolver(H, S)                       # :394
        ritz = V @ Y[:, :lowest]                                              # :397
        R = AV @ Y - (BV @ Y) * theta[None, :]                                # :401-410
        errors = np.array([norm(R[:, j]) for j in range(lowest)])             # :412-414
        if trace is not None:
            trace.widths.append(V.shape[1])
            trace.errors.append(errors)
        if (errors < tolerance).all():                                        # :416-419
            iters = i
            if trace is not None:
                trace.converged = True
            break
        if V.shape[1] <= max_dim:                                             # :422
            T = compute_DPR_free(theta, R, diag_matrix, diag_second_matrix)   # :428
            V = lapack_qr(concatenate(V, T))                                  # :431-434
        else:
            V = V @ Y[:, :initial_dimension]                                  # :438
    return theta[:lowest].copy(), np.asfortranarray(ritz), iters              # :451


def free_matmul(fun, array):
    """davidson.f90:526-569: out(i,j) = dot(fun(i, dim), array(:, j)), i 1-based."""
    array = np.asarray(array, dtype=np.float64)
    dim1 = array.shape[0]
    out = np.zeros(array.shape, order="F")
    for i in range(1, dim1 + 1):
        out[i - 1, :] = fun(i, dim1) @ array
    return out


# --------------------------------------------------------------------------------------------
# the matrix-free harness operators of the reference tests (tests/test_utils.f90:38-116,
# duplicated in benchmark_free.f90:38-76)
# --------------------------------------------------------------------------------------------
def harness_exp_table(dim):
    """e_i = exp(real(i)/real(dim)), i=1..dim: SINGLE precision exp of a single precision quotient,
    then widened to fp64 (tests/test_utils.f90:82,85)."""
    i = np.arange(1, dim + 1, dtype=np.float32)
    return np.exp(i / np.float32(dim), dtype=np.float32).astype(np.float64)


_SCALE = float(np.float32(1e-4))      # the `1e-4` literal is default real (tests/test_utils.f90:87)


def _harness_row(i, dim, trig, e=None):
    """tests/test_utils.f90:72-116: row/column i (1-based) of the off-diagonal generator:
    trig(log(sqrt(atan2(e_min_index, e_max_index)))) * 1e-4 where for j>=i the arguments are
    atan2(e_i, e_j) and for j<i atan2(e_j, e_i)."""
    if e is None:
        e = harness_exp_table(dim)
    x = e[i - 1]
    j = np.arange(1, dim + 1)
    first = np.where(j >= i, x, e)
    second = np.where(j >= i, e, x)
    return trig(np.log(np.sqrt(np.arctan2(first, second)))) * _SCALE


def compute_matrix_on_the_fly(i, dim, e=None):
    """tests/test_utils.f90:38-52: cos-generator, diagonal += real(i)."""
    v = _harness_row(i, dim, np.cos, e)
    v[i - 1] = v[i - 1] + float(np.float32(i))
    return v


def compute_stx_on_the_fly(i, dim, e=None):
    """tests/test_utils.f90:55-68: sin-generator, diagonal = 1."""
    v = _harness_row(i, dim, np.sin, e)
    v[i - 1] = 1.0
    return v


def harness_matrices(dim):
    """tests/test_free_numpy.f90:18-21: column j = row function(j)."""
    e = harness_exp_table(dim)
    mtx = np.zeros((dim, dim), order="F")
    stx = np.zeros((dim, dim), order="F")
    for j in range(1, dim + 1):
        mtx[:, j - 1] = compute_matrix_on_the_fly(j, dim, e)
        stx[:, j - 1] = compute_stx_on_the_fly(j, dim, e)
    return mtx, stx


def apply_mtx_to_vect(x):
    """tests/test_utils.f90:11-21 (free_matmul of the row function: row i of the applied matrix is
    compute_matrix_on_the_fly(i))."""
    dim = x.shape[0]
    e = harness_exp_table(dim)
    return free_matmul(lambda i, d: compute_matrix_on_the_fly(i, d, e), x)


def apply_stx_to_vect(x):
    """tests/test_utils.f90:23-33."""
    dim = x.shape[0]
    e = harness_exp_table(dim)
    return free_matmul(lambda i, d: compute_stx_on_the_fly(i, d, e), x)
