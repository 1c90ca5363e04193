"""Randomised sweep of the multi-rank engine (2-6 ranks as threads on one GPU, loopback transport of the TEST build) against the one-rank
engine: same iteration count on every rank, eigenvalues equal to 1e-11 - over orders that do not divide by the rank count or the tile
edge, both storages, dense / hashed operators, standard / generalized, DPR / GJD, with and without restarts.  Needs the test build
(DAVIDSON_HIP_LIB=fortran_davidson_amd/lib/test/libdavidson_hip.so, as tests/conftest.py sets it):
    python tests/multirank_parity_sweep.py [ncases] [seed]"""
import ctypes as C
import os
import sys
import threading
import time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
_T = os.path.join(ROOT, "fortran_davidson_amd", "lib", "test")
os.environ.setdefault("DAVIDSON_HIP_LIB", os.path.join(_T, "libdavidson_hip.so"))
os.environ["LD_LIBRARY_PATH"] = _T + ":" + os.environ.get("LD_LIBRARY_PATH", "")
import numpy as np
import torch  # noqa: F401
import fortran_davidson_amd as fd

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def solve(n, lowest, max_dim, gev, storage, kind, sp, seed, method, tol, nranks):
    engs = [fd.DavidsonEngine(n, lowest, max_dim, gev=gev, rank=r, nranks=nranks, storage=storage) for r in range(nranks)]
    if nranks > 1:
        handles = (C.c_void_p * nranks)(*[e.c.h for e in engs])
        assert fd.hip_lib().dav_local_group_join(handles, nranks) == 0
    out, err = [None] * nranks, [None] * nranks

    def work(r):
        try:
            e = engs[r]
            if kind == "dense":
                e.generate_diagonal_dominant(1, sp, seed=seed)
            else:
                e.set_hashed_operator(1, sp, seed=seed)
            if gev:
                e.set_hashed_operator(2, sp, 1.0, seed=seed + 1000) if kind == "hashed" else e.generate_diagonal_dominant(2, sp, 1.0, seed=seed + 1000)
            out[r] = e.solve(method, 80, tol, want_vectors=False)
        except Exception as exc:      # noqa: BLE001
            err[r] = exc

    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    for e in engs:
        e.close()
    assert all(x is None for x in err), err
    assert all(o is not None for o in out), "a rank did not finish"
    return out


bad = 0
t0 = time.time()
for case in range(ncases):
    n = int(rng.choice([300, 777, 1300, 2305, 4099, 6000]))
    lowest = int(rng.choice([1, 3, 4, 8]))
    sp = float(rng.choice([1e-3, 1e-2, 3e-2]))
    gev = bool(rng.integers(2))
    kind = ["dense", "hashed"][int(rng.integers(2))]
    method = "DPR" if kind == "hashed" else ["DPR", "GJD"][int(rng.integers(2))]
    max_dim = [None, 3 * lowest][int(rng.integers(2))]
    storage = ["full", "symmetric"][int(rng.integers(2))]
    nranks = int(rng.choice([2, 3, 4, 5, 6]))
    seed = int(rng.integers(1, 1000))
    tol = float(rng.choice([1e-6, 1e-8]))
    if kind == "hashed" and not gev:
        gev = True                      # matrix-free operators run the generalized driver (src/davidson.f90:277-460)
    one = solve(n, lowest, max_dim, gev, storage, kind, sp, seed, method, tol, 1)[0]
    many = solve(n, lowest, max_dim, gev, storage, kind, sp, seed, method, tol, nranks)
    ok = all(o[2] == one[2] and np.array_equal(o[0], many[0][0]) for o in many) and np.abs(many[0][0] - one[0]).max() < 1e-11 * max(1.0, np.abs(one[0]).max())
    bad += not ok
    print(f"n={n:5d} lowest={lowest} sparsity={sp:g} gev={int(gev)} {kind:6s} {method} max_dim={max_dim} storage={storage:9s} ranks={nranks} tol={tol:g}: "
          f"iters {one[2]} / {[o[2] for o in many]}, |dlam| {np.abs(many[0][0] - one[0]).max():.1e}{'' if ok else '   <-- MISMATCH'}", flush=True)
print(f"{ncases} cases in {time.time() - t0:.0f} s, mismatches: {bad}")
