"""Checker tool (not collected by pytest; uses the oracle): the drop-in dense call against the oracle on STRUCTURED matrices - the
classes the generator-based sweeps never produce: banded, block diagonal, sparse, diagonals that do not ascend with the index,
repeated diagonal entries, negative and scaled spectra, strong coupling.  Prints one line per case; a case counts as a mismatch
when the engine does not converge where the oracle does, its eigenvalues differ by more than 1e-7 or its residuals exceed the
tolerance; differing iteration counts are reported separately (completion directions of rank-deficient blocks are arbitrary in the
reference too).      python tests/structured_parity_sweep.py [ncases] [seed] [only]
`only` = comma-separated case numbers (0-based): every case is still DRAWN - the random stream is the sweep's - but only those are
solved: how tests/golden/structured_iteration_differences.json re-runs single cases of a logged sweep."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import fortran_davidson_amd as fd          # noqa: E402
from oracle import davidson_oracle as O    # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = {int(x) for x in sys.argv[3].split(",")} if len(sys.argv) > 3 else None


def sym(a):
    return 0.5 * (a + a.T)


def make(kind, n):
    d = 1.0 + np.arange(n, dtype=np.float64)
    if kind == "banded":
        bw = int(rng.integers(1, 6))
        eps = float(rng.choice([1e-3, 1e-2, 0.1, 0.3]))
        a = np.diag(d)
        for off in range(1, bw + 1):
            a += eps / off * (np.eye(n, k=off) + np.eye(n, k=-off))
        return a, f"bw={bw} eps={eps}"
    if kind == "blockdiag":
        nb = int(rng.choice([2, 3, 5]))
        a = np.diag(d)
        edges = np.linspace(0, n, nb + 1).astype(int)
        for s, e in zip(edges[:-1], edges[1:]):
            h = rng.random((e - s, e - s)) * 1e-2
            a[s:e, s:e] += sym(h) - np.diag(np.diag(sym(h)))
        return a, f"blocks={nb}"
    if kind == "sparse":
        dens = float(rng.choice([0.002, 0.01, 0.05]))
        mask = rng.random((n, n)) < dens
        h = np.where(mask, rng.random((n, n)) * 0.05, 0.0)
        a = np.diag(d) + sym(h) - np.diag(np.diag(sym(h)))
        return a, f"density={dens}"
    if kind == "permuted":
        p = rng.permutation(n)
        h = rng.random((n, n)) * 1e-3
        a = np.diag(d[p]) + sym(h) - np.diag(np.diag(sym(h)))
        return a, "diag permuted"
    if kind == "ties":
        dd = np.floor(d / 2.0) + 1.0                     # every diagonal value twice
        h = rng.random((n, n)) * 1e-3
        a = np.diag(dd) + sym(h) - np.diag(np.diag(sym(h)))
        return a, "diag in pairs"
    if kind == "negative":
        h = rng.random((n, n)) * 1e-3
        a = np.diag(-d[::-1] * 0.5) + sym(h) - np.diag(np.diag(sym(h)))
        return a, "negative spectrum"
    if kind == "scaled":
        sc = float(rng.choice([1e-6, 1e6]))
        h = rng.random((n, n)) * 1e-3
        a = (np.diag(d) + sym(h) - np.diag(np.diag(sym(h)))) * sc
        return a, f"scale={sc}"
    h = rng.random((n, n)) * 0.2                         # "strong": far from diagonal dominance
    a = np.diag(d) + sym(h) - np.diag(np.diag(sym(h)))
    return a, "strong coupling"


kinds = ["banded", "blockdiag", "sparse", "permuted", "ties", "negative", "scaled", "strong"]
t0 = time.time()
bad = differ = 0
for case in range(ncases):
    kind = kinds[case % len(kinds)]
    n = int(rng.choice([120, 300, 513, 800]))
    lowest = int(rng.choice([1, 3, 4, 8]))
    method = "GJD" if rng.random() < 0.25 else "DPR"
    gev = rng.random() < 0.3
    storage = str(rng.choice(["full", "symmetric"]))
    a, what = make(kind, n)
    scale = np.abs(np.diag(a)).max() / n
    tol = 1e-8 * max(scale, 1e-300) if kind == "scaled" else 1e-8
    b = None
    if gev:
        # second operator: near the identity (the reference's tests), or far from it - diagonal spread over one or two decades, or
        # on two levels (1 and 0.01: the start vectors, chosen by the diagonal of A alone, then miss the lowest pairs)
        bkind = int(rng.integers(0, 4))
        db = [np.ones(n), 1.0 + 9.0 * rng.random(n), np.linspace(1.0, 100.0, n), np.where(rng.random(n) < 0.5, 1.0, 1e-2)][bkind]
        hb = rng.random((n, n)) * 1e-3 * db.min()
        b = np.diag(db) + sym(hb) - np.diag(np.diag(sym(hb)))
        what += f" B{bkind}"
    # restart width: the default (10 * lowest) or a narrow one (restarts between the completions of rank-deficient blocks)
    md = None if rng.random() < 0.6 else int(rng.integers(2 * lowest, 5 * lowest + 1))
    what += "" if md is None else f" md={md}"
    if only is not None and case not in only:
        continue
    os.environ["DAVIDSON_STORAGE"] = storage
    try:
        with np.errstate(all="ignore"):
            lam_o, _, it_o = O.generalized_eigensolver_dense(a, lowest, method, 200, tol, md, b)
    except RuntimeError as exc:          # the reference's `error stop` after a failed LAPACK call (NaN in the projected matrix)
        lam_o, it_o = np.full(lowest, np.nan), 999
        what += " [oracle: " + str(exc)[-30:] + "]"
    lam, vec, it = fd.generalized_eigensolver(a, lowest, method, 200, tol, md, b)
    bx = vec if b is None else b @ vec
    res = np.linalg.norm(a @ vec - bx * lam[None, :], axis=0).max()
    ok_o = it_o <= 200 and np.isfinite(lam_o).all()
    ok_e = it <= 200 and np.isfinite(lam).all() and res < 10 * tol
    dl = np.abs(lam - lam_o).max() / max(scale, 1e-300) if ok_o and np.isfinite(lam).all() else float("nan")
    flag = ""
    if ok_o and ok_e and not dl < 1e-7:
        # both converged, to different pairs: the true spectrum decides (the reference's loop can settle on a pair that is not the
        # lowest when its start vectors miss it)
        import scipy.linalg as sl
        truth = sl.eigh(a, b, eigvals_only=True)[:lowest] if b is not None else np.linalg.eigvalsh(a)[:lowest]
        if np.abs(lam - truth).max() / max(scale, 1e-300) < 1e-7:
            flag = "   (the engine has the lowest pairs, the oracle does not)"
        else:
            flag = "   <-- MISMATCH"
            bad += 1
    elif ok_o and not ok_e:
        flag = "   <-- MISMATCH"
        bad += 1
    elif ok_o and it != it_o:
        flag = "   (iterations differ)"
        differ += 1
    elif not ok_o and ok_e:
        flag = "   (engine converged, oracle did not)"
    print(f"#{case:<4d}" * (only is not None) + f"{kind:9s} {what:18s} n={n:4d} lowest={lowest} {method} gev={int(gev)} {storage:9s}: oracle iters {it_o:3d}, engine {it:3d}, "
          f"|dlam|/scale {dl:.1e}, residual {res:.1e}{flag}", flush=True)
print(f"{ncases} cases in {time.time() - t0:.0f} s, mismatches: {bad}, iteration counts differ: {differ}")
