"""Sustained load at full size: configs[2] (N=200000, lowest=16, DPR) solved 300 times and configs[3] (generalized, GJD) 60 times on one
engine each - every solve must return bitwise the same eigenvalues and iteration count; the per-solve time over the run shows what the
chip holds when it is warm.
    python profiles/tools/soak_full_size.py"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch  # noqa: F401
import fortran_davidson_amd as fd

for name, n, lowest, gev, method, reps in (("configs[2]", 200000, 16, False, "DPR", 300), ("configs[3]", 200000, 8, True, "GJD", 60)):
    with fd.DavidsonEngine(n, lowest, 80, gev=gev, storage="symmetric") as eng:
        eng.generate_diagonal_dominant(1, 1e-3, seed=1)
        if gev:
            eng.set_hashed_operator(2, 1e-3, 1.0, seed=2)
        lam0, _, it0 = eng.solve(method, 1000, 1e-8, want_vectors=False)
        times = []
        for r in range(reps):
            eng.c.synchronize()
            t0 = time.perf_counter()
            lam, _, it = eng.solve(method, 1000, 1e-8, want_vectors=False)
            eng.c.synchronize()
            times.append(time.perf_counter() - t0)
            assert it == it0 and np.array_equal(lam, lam0), (r, it, it0)
        t = np.array(times) * 1e3
        k = max(reps // 10, 1)
        print(f"{name}: {reps} solves, iters {it0}, bitwise identical; ms per solve: first {k} {t[:k].mean():.2f}, last {k} {t[-k:].mean():.2f}, "
              f"min {t.min():.2f}, median {np.median(t):.2f}, max {t.max():.2f}", flush=True)
