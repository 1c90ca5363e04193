"""Outer iteration counts of the GJD solve against the oracle (the reference's exact DSYSV solves, oracle/davidson_oracle.py) over a
grid of problems, for several settings of the inner tolerances (DAV_GJD_ADAPTIVE: wanted pairs, DAV_GJD_TOL_UNWANTED: the others):
    python tests/gjd_policy_sweep.py
Checker tool: it uses the oracle, which is test infrastructure, so it lives under tests/ (not collected by pytest; the suite runs a small
version of it: test_solver_gpu.py::test_gjd_outer_iterations_equal_the_oracles_exact_solves_on_a_grid)."""
import itertools
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch  # noqa: F401
import fortran_davidson_amd as fd
from oracle import davidson_oracle as O

# a negative tolerance marks the unwanted pairs as followers: they also stop when every wanted pair has stopped
settings = [("0", "1e-4"), ("0.01", "1e-2"), ("0.01", "-1e-2"), ("0.01", "-1e-4"), ("0.01", "1e-1"), ("0.01", "0.5")]
orders = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "150,300,500".split(","))]
tol = float(os.environ.get("SWEEP_TOL", "1e-8"))
bad = {s: [] for s in settings}
ncases = 0
t0 = time.time()
for n, lowest, sp, gev, seed in itertools.product(orders, (2, 4, 8), (1e-3, 1e-2, 5e-2), (False, True), (1, 2)):
    A = O.generate_diagonal_dominant(n, sp, seed=seed)
    B = O.generate_diagonal_dominant(n, sp, 1.0, seed=seed + 100) if gev else None
    lam_o, _, it_o = O.generalized_eigensolver_dense(A, lowest, "GJD", 50, tol, None, B)
    ncases += 1
    line = f"n={n:4d} lowest={lowest} sparsity={sp:g} gev={int(gev)} seed={seed}: oracle iters {it_o:2d} |"
    for s in settings:
        os.environ["DAV_GJD_ADAPTIVE"], os.environ["DAV_GJD_TOL_UNWANTED"] = s
        lam, vec, it = fd.generalized_eigensolver(A, lowest, "GJD", 50, tol, None, B)
        BX = vec if B is None else B @ vec
        res = np.linalg.norm(A @ vec - BX * lam[None, :], axis=0).max()
        ok = it == it_o and np.abs(lam - lam_o).max() < 1e-8 and (res < tol or it > 50)
        if not ok:
            bad[s].append((n, lowest, sp, gev, seed, it_o, it, float(res)))
        line += f" {it:2d}{'' if ok else '!'}"
    print(line, flush=True)
print(f"{ncases} cases in {time.time() - t0:.0f} s; settings (DAV_GJD_ADAPTIVE, DAV_GJD_TOL_UNWANTED) and their mismatches:")
for s in settings:
    print(" ", s, len(bad[s]), bad[s][:6])
