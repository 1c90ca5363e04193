"""TEST INFRASTRUCTURE ONLY - generates tests/golden/* by running the REFERENCE ITSELF
(oracle/_ref, built from /root/reference by oracle/build_ref.sh) in this container.

    python -m oracle.gen_golden          # from the repo root, where /root/reference exists

Fixtures are data only: inputs (a seed for our counter-based generator, or the reference's own test
data file src/tests/matrix.txt as numbers) and the outputs the reference produced for them
(eigenvalues, eigenvectors, iteration counts), plus scipy `eigh` values as the reference's Python
checkers compute them (src/tests/test_davidson.py:36-40,67-69).
"""
from __future__ import annotations

import json
import os

import numpy as np
import scipy.linalg

from oracle import davidson_oracle as O
from oracle import ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
REF_MATRIX_TXT = "/root/reference/src/tests/matrix.txt"

# (name, n, lowest, sparsity, seedA, gev(seedB) or None, method, max_dim, tol)
DENSE_CASES = [
    ("c1_n50_std_dpr", 50, 3, 1e-3, 1, None, "DPR", None, 1e-8),
    ("c1_n50_std_gjd", 50, 3, 1e-3, 1, None, "GJD", None, 1e-8),
    ("c1_n50_gev_dpr", 50, 3, 1e-3, 1, 2, "DPR", 10, 1e-8),
    ("c1_n50_gev_gjd", 50, 3, 1e-3, 1, 2, "GJD", 10, 1e-8),
    ("n100_main_gev_dpr", 100, 3, 1e-3, 3, 4, "DPR", 10, 1e-5),     # shape of src/main.f90:44-54
    ("n100_main_gev_gjd", 100, 3, 1e-3, 3, 4, "GJD", 10, 1e-5),
    ("n400_std_dpr", 400, 3, 1e-3, 1, None, "DPR", None, 1e-8),
    ("n400_std_gjd", 400, 3, 1e-3, 1, None, "GJD", None, 1e-8),
    ("n400_gev_gjd", 400, 3, 1e-3, 1, 2, "GJD", None, 1e-8),
    ("n1000_restart_dpr", 1000, 4, 5e-2, 1, None, "DPR", None, 1e-8),
    ("n1000_gev_restart_dpr", 1000, 4, 2e-2, 1, 2, "DPR", None, 1e-8),
    ("n2000_std_dpr", 2000, 8, 1e-3, 1, None, "DPR", None, 1e-8),
    ("n3000_hard_dpr", 3000, 8, 2e-2, 5, None, "DPR", None, 1e-8),
    ("n4000_gev_dpr", 4000, 8, 1e-3, 1, 2, "DPR", None, 1e-8),
]
FREE_CASES = [("free_n50", 50, 3, 20), ("free_n300", 300, 3, 20)]


def _resid(A, B, lam, X):
    BX = X if B is None else B @ X
    return np.linalg.norm(A @ X - BX * lam[None, :], axis=0)


def main():
    os.makedirs(OUT, exist_ok=True)
    arrays = {}
    manifest = {"dense": {}, "free": {}, "generator": {}, "lapack": {}}

    # ---- generator pin: a corner block + checksums so host/device generators can be checked
    A = O.generate_diagonal_dominant(64, 1e-3, seed=1)
    arrays["gen__block64_seed1"] = A
    B = O.generate_diagonal_dominant(64, 1e-3, 1.0, seed=2)
    arrays["gen__block64_seed2_diag1"] = B
    manifest["generator"] = {"n": 64, "sparsity": 1e-3,
                             "sum_seed1": float(A.sum()), "sum_seed2": float(B.sum())}

    # ---- the reference's own data file (src/tests/matrix.txt, 100x100 row-major text)
    M = np.loadtxt(REF_MATRIX_TXT).reshape(100, 100)
    arrays["matrix_txt__A"] = np.asfortranarray(M)
    for meth in ("DPR", "GJD"):
        lam, vec, it = ref.dense_solve(M, 3, meth, 1000, 1e-8)
        name = f"matrix_txt_{meth.lower()}"
        arrays[f"{name}__evals"] = lam
        arrays[f"{name}__evecs"] = vec
        manifest["dense"][name] = dict(matrix="matrix_txt", n=100, lowest=3, method=meth, max_it=1000,
                                       tol=1e-8, max_dim=None, gev=False, iters=int(it),
                                       resid=_resid(M, None, lam, vec).tolist(),
                                       eigh=scipy.linalg.eigh(M, eigvals_only=True)[:3].tolist())

    # ---- generator matrices through the reference
    for name, n, L, sp, sa, sb, meth, md, tol in DENSE_CASES:
        A = O.generate_diagonal_dominant(n, sp, seed=sa)
        B = O.generate_diagonal_dominant(n, sp, 1.0, seed=sb) if sb is not None else None
        lam, vec, it = ref.dense_solve(A, L, meth, 1000, tol, md, B)
        tr = O.Trace()
        lam_o, _, it_o = O.generalized_eigensolver_dense(A, L, meth, 1000, tol, md, B, trace=tr)
        assert it_o == it and np.abs(lam_o - lam).max() < 1e-10, (name, it, it_o)
        arrays[f"{name}__evals"] = lam
        if n <= 400:
            arrays[f"{name}__evecs"] = vec
        eigh = scipy.linalg.eigh(A, B, eigvals_only=True, subset_by_index=[0, L - 1])
        manifest["dense"][name] = dict(matrix="generator", n=n, lowest=L, sparsity=sp, seed_a=sa, seed_b=sb,
                                       method=meth, max_it=1000, tol=tol, max_dim=md, gev=sb is not None,
                                       iters=int(it), widths=tr.widths,
                                       resid=_resid(A, B, lam, vec).tolist(), eigh=eigh.tolist())
        print(name, it, tr.widths, lam[:3])

    # ---- matrix-free harness (tests/test_utils.f90 operators) through the reference
    for name, n, L, md in FREE_CASES:
        lam, vec, it = ref.free_solve_harness(n, L, 1000, 1e-8, md)
        mtx, stx = ref.harness_matrices(n)
        arrays[f"{name}__evals"] = lam
        arrays[f"{name}__evecs"] = vec
        if n <= 50:
            arrays[f"{name}__mtx"] = mtx
            arrays[f"{name}__stx"] = stx
        eigh = scipy.linalg.eigh(mtx, stx, eigvals_only=True, subset_by_index=[0, L - 1])
        manifest["free"][name] = dict(n=n, lowest=L, max_dim=md, tol=1e-8, max_it=1000, iters=int(it),
                                      resid=_resid(mtx, stx, lam, vec).tolist(), eigh=eigh.tolist())
        print(name, it, lam)

    # ---- wrapper level (mirrors src/tests/test_call_lapack.f90): DSYEV, DSYGV, QR on 50x50
    A = O.generate_diagonal_dominant(50, 1e-3, seed=7)
    B = O.generate_diagonal_dominant(50, 1e-3, 1.0, seed=8)
    w, v = ref.lapack_eigensolver(A)
    arrays["lapack__dsyev_w"], arrays["lapack__dsyev_v"] = w, v
    w, v = ref.lapack_eigensolver(A, B)
    arrays["lapack__dsygv_w"], arrays["lapack__dsygv_v"] = w, v
    arrays["lapack__qr_q"] = ref.lapack_qr(A[:, :20])
    arrays["lapack__precond"] = ref.generate_preconditioner(np.diag(A)[::-1].copy(), 6)
    manifest["lapack"] = dict(n=50, seed_a=7, seed_b=8, sparsity=1e-3, qr_cols=20)

    np.savez_compressed(os.path.join(OUT, "reference_cases.npz"), **arrays)
    with open(os.path.join(OUT, "reference_cases.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
