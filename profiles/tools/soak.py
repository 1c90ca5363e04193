"""Repeated solves and engine life cycles on one GPU: device memory before / after (leak check), results identical every time.
    python profiles/tools/soak.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch
import fortran_davidson_amd as fd


def free_mb():
    fd.free_buffers()          # what the buffer cache keeps of destroyed engines is not a leak (dav_free_buffers)
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20


torch.cuda.init()
base = free_mb()
print(f"free at start {base:.0f} MiB", flush=True)
# 1. one engine, many solves (DPR and GJD, standard and generalized)
for gev, method, n, lowest in [(False, "DPR", 20000, 8), (True, "GJD", 6000, 4), (False, "GJD", 6000, 8)]:
    with fd.DavidsonEngine(n, lowest, None, gev=gev, storage="symmetric") as eng:
        eng.generate_diagonal_dominant(1, 1e-3, seed=1)
        if gev:
            eng.set_hashed_operator(2, 1e-3, 1.0, seed=2)
        lam0, _, it0 = eng.solve(method, 1000, 1e-8, want_vectors=False)
        f0 = free_mb()
        for _ in range(300):
            lam, _, it = eng.solve(method, 1000, 1e-8, want_vectors=False)
            assert it == it0 and np.array_equal(lam, lam0)
        f1 = free_mb()
        print(f"{method} gev={gev} n={n}: 300 solves, iters {it0}, free {f0:.0f} -> {f1:.0f} MiB (delta {f1 - f0:+.0f})", flush=True)
        assert abs(f1 - f0) < 64
print(f"after the engines closed: free {free_mb():.0f} MiB (start {base:.0f})", flush=True)
# 2. engine life cycles through the drop-in entry (upload, solve, free)
A = np.asfortranarray(np.random.default_rng(0).standard_normal((1500, 1500))); A = A + A.T + np.diag(np.arange(1500) * 10.0)
f0 = free_mb()
for _ in range(60):
    lam, vec, it = fd.generalized_eigensolver(A, 4, "DPR", 200, 1e-8)
f1 = free_mb()
print(f"60 drop-in calls (create, upload, solve, destroy): free {f0:.0f} -> {f1:.0f} MiB (delta {f1 - f0:+.0f})", flush=True)
assert abs(f1 - f0) < 64 and abs(free_mb() - base) < 256
print("OK")
