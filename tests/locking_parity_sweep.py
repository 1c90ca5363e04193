"""Randomised parity sweep of the opt-in "locking" policy against its oracle statement (oracle/davidson_oracle.py:
generalized_eigensolver_dense_locking): iteration counts exactly, eigenvalues to 1e-8, residuals below the tolerance - over orders,
numbers of wanted pairs, couplings, restart widths, both storages, DPR and (small orders) GJD; half of the matrices get a clustered
lowest diagonal so that the pairs lock at different iterations.  Checker tool (uses the oracle: lives under tests/, not collected by
pytest; a fixed-seed slice runs in tests/test_parity_sweeps_gpu.py):
    python tests/locking_parity_sweep.py [ncases] [seed] [only]
`only` = comma-separated case numbers (the #n of a printed line): every case is still drawn, only those are solved."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch  # noqa: F401
import fortran_davidson_amd as fd
from oracle import davidson_oracle as O

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = {int(x) for x in sys.argv[3].split(",")} if len(sys.argv) > 3 else None
bad = 0
done = 0
t0 = time.time()
for case in range(ncases):
    method = "GJD" if rng.integers(4) == 0 else "DPR"
    n = int(rng.choice([150, 300, 500] if method == "GJD" else [200, 400, 777, 1024, 1500, 2305]))
    lowest = int(rng.choice([1, 2, 3, 5, 8, 12]))
    sp = float(rng.choice([1e-3, 1e-2, 3e-2, 5e-2]))
    max_dim = [None, 3 * lowest, 4 * lowest, 6 * lowest][int(rng.integers(4))]
    storage = ["full", "symmetric"][int(rng.integers(2))]
    seed = int(rng.integers(1, 1000))
    tol = float(rng.choice([1e-6, 1e-8]))
    if 3 * lowest > n // 4:
        continue
    A = O.generate_diagonal_dominant(n, sp, seed=seed)
    if rng.integers(2):
        d = np.arange(1, n + 1, dtype=float) + 2.0
        d[:lowest + 2] = np.sort(1.0 + rng.random(lowest + 2) * np.array([0.2 if i % 3 else 6.0 for i in range(lowest + 2)]).cumsum())
        A[np.arange(n), np.arange(n)] = d
    # a third of the problems generalized (round 6): a second operator near the identity, or with a diagonal spread over a decade
    gev = rng.integers(3) == 0
    B = None
    if gev:
        B = O.generate_diagonal_dominant(n, sp, 1.0, seed=seed + 1)
        if rng.integers(2):
            B[np.arange(n), np.arange(n)] = 1.0 + 9.0 * rng.random(n)
    if only is not None and case not in only:
        continue
    if os.environ.get("SWEEP_DRY"):                # list the cases without solving them (to find the #n of a logged line)
        print(f"#{case:<4d}{method} gev={int(gev)} n={n:5d} lowest={lowest:2d} sparsity={sp:g} max_dim={max_dim} storage={storage:9s} tol={tol:g} seed={seed:3d}")
        continue
    lam_o, vec_o, it_o = O.generalized_eigensolver_dense_locking(A, lowest, method, 80, tol, max_dim, second_matrix=B)
    with fd.DavidsonEngine(n, lowest, max_dim, gev=gev, storage=storage) as eng:
        eng.set_correction_policy("locking")
        eng.set_dense(1, A)
        if gev:
            eng.set_dense(2, B)
        lam, vec, it = eng.solve(method, 80, tol)
    res = np.linalg.norm(A @ vec - (vec if B is None else B @ vec) * lam[None, :], axis=0).max()
    conv = it_o <= 80
    # (GJD: the engine's inner solves are inexact by default - never MORE outer iterations than the oracle's exact solves)
    ok = (it == it_o if method == "DPR" else it <= it_o) and np.abs(lam - lam_o).max() < 1e-8 * max(1.0, np.abs(lam_o).max()) and (res < tol or not conv)
    done += 1
    bad += not ok
    print(f"#{case:<4d}{method} gev={int(gev)} n={n:5d} lowest={lowest:2d} sparsity={sp:g} max_dim={max_dim} storage={storage:9s} tol={tol:g} seed={seed:3d}: "
          f"oracle iters {it_o:2d}, engine {it:2d}, |dlam| {np.abs(lam - lam_o).max():.1e}, residual {res:.1e}{'' if ok else '   <-- MISMATCH'}", flush=True)
print(f"{done} cases in {time.time() - t0:.0f} s, mismatches: {bad}")
