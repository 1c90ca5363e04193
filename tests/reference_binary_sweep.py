"""Fresh random problems solved by the REFERENCE ITSELF (oracle/_ref: /root/reference's sources compiled with flang + MKL by
oracle/build_ref.sh; the .so travels to the GPU box) and by the engine: iteration counts, eigenvalues, residuals.  DPR and GJD, standard and
generalized.  The reference runs in a CHILD process (its threaded MKL and the product's sequential MKL + PyTorch's OpenMP runtime do not
share one process - bench.py's cpu_baseline leg does the same).  The golden fixtures under tests/golden/ are the pinned subset of this;
checker tool (uses oracle/: lives under tests/):
    python tests/reference_binary_sweep.py [ncases] [seed]"""
import json
import os
import subprocess
import sys
import time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np


def cases(ncases, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(ncases):
        method = ["DPR", "GJD"][int(rng.integers(2))]
        n = int(rng.choice([200, 400, 700, 1000] if method == "GJD" else [300, 1000, 2000, 4000]))
        lowest = int(rng.choice([1, 3, 4, 8]))
        sp = float(rng.choice([1e-4, 1e-3, 1e-2, 3e-2]))
        gev = bool(rng.integers(2))
        max_dim = [None, 3 * lowest, 6 * lowest][int(rng.integers(3))]
        out.append((method, n, lowest, sp, gev, max_dim, int(rng.integers(1, 1000))))
    return out


def free_cases(ncases, seed):
    rng = np.random.default_rng(seed + 77)
    out = []
    for _ in range(ncases):
        lowest = int(rng.choice([1, 3, 4, 8]))
        out.append((int(rng.choice([150, 300, 600, 1000])), lowest, float(rng.choice([1e-4, 1e-3, 1e-2])), bool(rng.integers(2)),
                    [2 * lowest, 4 * lowest, 10 * lowest][int(rng.integers(3))], int(rng.integers(1, 1000))))
    return out


def matrices(n, sp, gev, seed):
    from oracle import davidson_oracle as O
    A = O.generate_diagonal_dominant(n, sp, seed=seed)
    B = O.generate_diagonal_dominant(n, sp, 1.0, seed=seed + 1000) if gev else None
    return A, B


ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if len(sys.argv) > 3 and sys.argv[3] == "--reference-child":
    from oracle import ref
    if not ref.available():
        raise SystemExit("oracle/_ref is not built (run __graft_entry__.build() where /root/reference exists)")
    res = []
    for method, n, lowest, sp, gev, max_dim, s in cases(ncases, seed):
        A, B = matrices(n, sp, gev, s)
        lam, _, it = ref.dense_solve(A, lowest, method, 60, 1e-8, max_dim, B)
        res.append({"lam": [float(x) for x in lam], "iters": int(it)})
    # the matrix-free driver (src/davidson.f90:277-460) with numpy callbacks on the same kind of matrices (B hashed or the identity)
    free = []
    for n, lowest, sp, identity_b, max_dim, s in free_cases(ncases // 4, seed):
        A, B = matrices(n, sp, True, s)
        if identity_b:
            B = np.eye(n)
        lam, _, it = ref.free_solve_callbacks(n, lambda x: A @ x, lambda x: B @ x, lowest, 60, 1e-8, max_dim)
        free.append({"lam": [float(x) for x in lam], "iters": int(it)})
    print("REFERENCE_RESULTS " + json.dumps({"dense": res, "free": free}))
    raise SystemExit(0)

t0 = time.time()
def big_stack():
    # the reference keeps four N x N temporaries of its GJD correction on the stack (src/davidson.f90:700-734: automatic arrays):
    # with the default 8 MB stack it ends in a segmentation fault from N ~ 500 on
    import resource
    hard = resource.getrlimit(resource.RLIMIT_STACK)[1]
    resource.setrlimit(resource.RLIMIT_STACK, (hard, hard))


env = dict(os.environ)
ncpu = len(os.sched_getaffinity(0))
try:                                                     # the job's real CPU share (cgroup quota), as bench.py's baseline child uses it
    quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
    if quota != "max":
        ncpu = max(1, min(ncpu, int(int(quota) / int(period))))
except (OSError, ValueError):
    pass
env["OMP_NUM_THREADS"] = env["MKL_NUM_THREADS"] = str(ncpu)
env["HIP_VISIBLE_DEVICES"] = ""
child = subprocess.run([sys.executable, os.path.abspath(__file__), str(ncases), str(seed), "--reference-child"], capture_output=True, text=True,
                       timeout=3000, cwd=ROOT, env=env, preexec_fn=big_stack)
line = [ln for ln in child.stdout.splitlines() if ln.startswith("REFERENCE_RESULTS ")]
if child.returncode != 0 or not line:
    raise SystemExit("the reference child failed: " + (child.stdout + child.stderr)[-2000:])
both = json.loads(line[0][len("REFERENCE_RESULTS "):])
reference, reference_free = both["dense"], both["free"]
print(f"reference: {ncases} solves in {time.time() - t0:.0f} s", flush=True)
import torch  # noqa: F401,E402
import fortran_davidson_amd as fd  # noqa: E402
bad = 0
for (method, n, lowest, sp, gev, max_dim, s), r in zip(cases(ncases, seed), reference):
    A, B = matrices(n, sp, gev, s)
    lam, vec, it = fd.generalized_eigensolver(A, lowest, method, 60, 1e-8, max_dim, B)
    BX = vec if B is None else B @ vec
    res = np.linalg.norm(A @ vec - BX * lam[None, :], axis=0).max()
    lam_r = np.array(r["lam"])
    ok = it == r["iters"] and np.abs(lam - lam_r).max() < 1e-8 and (res < 1e-8 or r["iters"] > 60)
    bad += not ok
    print(f"{method} n={n:5d} lowest={lowest} sparsity={sp:g} gev={int(gev)} max_dim={max_dim} seed={s:3d}: reference iters {r['iters']:2d}, engine {it:2d}, "
          f"|dlam| {np.abs(lam - lam_r).max():.1e}, residual {res:.1e}{'' if ok else '   <-- MISMATCH'}", flush=True)
for (n, lowest, sp, identity_b, max_dim, s), r in zip(free_cases(ncases // 4, seed), reference_free):
    A, B = matrices(n, sp, True, s)
    if identity_b:
        B = np.eye(n)
    with fd.DavidsonEngine(n, lowest, max_dim, gev=True) as eng:
        eng.set_hashed_operator(1, sp, seed=s)
        eng.set_identity(2) if identity_b else eng.set_hashed_operator(2, sp, 1.0, seed=s + 1000)
        lam, vec, it = eng.solve("DPR", 60, 1e-8)
    res = np.linalg.norm(A @ vec - (B @ vec) * lam[None, :], axis=0).max()
    lam_r = np.array(r["lam"])
    ok = it == r["iters"] and np.abs(lam - lam_r).max() < 1e-8 and (res < 1e-8 or r["iters"] > 60)
    bad += not ok
    print(f"matrix-free n={n:5d} lowest={lowest} sparsity={sp:g} B={'I' if identity_b else 'hashed'} max_dim={max_dim} seed={s:3d}: reference iters "
          f"{r['iters']:2d}, engine {it:2d}, |dlam| {np.abs(lam - lam_r).max():.1e}, residual {res:.1e}{'' if ok else '   <-- MISMATCH'}", flush=True)
print(f"{ncases} dense + {ncases // 4} matrix-free cases in {time.time() - t0:.0f} s, mismatches: {bad}")
