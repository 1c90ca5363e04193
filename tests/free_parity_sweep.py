"""Randomised parity sweep of the MATRIX-FREE solves (device operators: the hashed diagonal-dominant generator for A, the same with unit
diagonal or the identity for B; row slabs or symmetric generation) against the oracle's restatement of generalized_eigensolver_free
(src/davidson.f90:277-460: always generalized, always DPR, non-sticky convergence test): iteration counts exactly, eigenvalues to 1e-8,
residuals below the tolerance.  Checker tool (uses the oracle: lives under tests/, not collected by pytest):
    python tests/free_parity_sweep.py [ncases] [seed]"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch  # noqa: F401
import fortran_davidson_amd as fd
from oracle import davidson_oracle as O

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
done = 0
t0 = time.time()
for case in range(ncases):
    n = int(rng.choice([150, 300, 513, 1000, 2048, 3000]))
    lowest = int(rng.choice([1, 2, 3, 5, 8]))
    sp = float(rng.choice([1e-4, 1e-3, 1e-2, 3e-2]))
    identity_b = bool(rng.integers(2))
    max_dim = [None, 2 * lowest, 4 * lowest][int(rng.integers(3))]
    storage = ["full", "symmetric"][int(rng.integers(2))]
    seed = int(rng.integers(1, 1000))
    tol = float(rng.choice([1e-6, 1e-8]))
    A = O.generate_diagonal_dominant(n, sp, seed=seed)
    B = np.eye(n) if identity_b else O.generate_diagonal_dominant(n, sp, 1.0, seed=seed + 1000)
    try:
        lam_o, vec_o, it_o = O.generalized_eigensolver_free(lambda x: A @ x, n, lowest, 60, tol, max_dim, lambda x: B @ x,
                                                            diag_matrix=np.diag(A).copy(), diag_second_matrix=np.diag(B).copy())
    except RuntimeError:
        continue
    with fd.DavidsonEngine(n, lowest, max_dim, gev=True, storage=storage) as eng:
        eng.set_hashed_operator(1, sp, seed=seed)
        if identity_b:
            eng.set_identity(2)
        else:
            eng.set_hashed_operator(2, sp, 1.0, seed=seed + 1000)
        lam, vec, it = eng.solve("DPR", 60, tol)
    res = np.linalg.norm(A @ vec - (B @ vec) * lam[None, :], axis=0).max()
    ok = it == it_o and np.abs(lam - lam_o).max() < 1e-8 * max(1.0, np.abs(lam_o).max()) and (res < tol or it_o > 60)
    done += 1
    bad += not ok
    print(f"n={n:5d} lowest={lowest} sparsity={sp:g} B={'I' if identity_b else 'hashed'} max_dim={max_dim} storage={storage:9s} tol={tol:g} seed={seed:3d}: "
          f"oracle iters {it_o:2d}, engine {it:2d}, |dlam| {np.abs(lam - lam_o).max():.1e}, residual {res:.1e}{'' if ok else '   <-- MISMATCH'}", flush=True)
print(f"{done} cases in {time.time() - t0:.0f} s, mismatches: {bad}")
