"""Randomised check of the kernels behind the C ABI against numpy: block sweeps W = Op X (dense host / device-generated / hashed /
the reference's matrix-free test operators in both forms, full rows and symmetric tiles with the one-, two- and four-block-row schedules forced at any order, 1-64 columns at random
offsets), Gram blocks P^T Q and panel products P M at random shapes and offsets.  Matrices come from the oracle's generator
(test infrastructure: this tool lives under tests/, not collected by pytest):
    python tests/kernel_fuzz.py [ncases] [seed]"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch  # noqa: F401
import fortran_davidson_amd as fd
from fortran_davidson_amd.engine_c import OP_A, PANEL_V, PANEL_W, PANEL_X, PANEL_S
from oracle import davidson_oracle as O

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


for case in range(ncases):
    n = int(rng.choice([17, 64, 255, 256, 257, 300, 511, 777, 1024, 1300, 2049, 2305, 3333, 5000]))
    storage = int(rng.integers(2))
    sched = str(rng.choice(["0", "1", "2", "4"])) if storage == 1 else "0"
    kind = str(rng.choice(["host", "generated", "hashed", "harness", "harness_sin"]))
    os.environ["DAV_SYM_R"] = sched
    os.environ["DAV_SYM_GEN_WIDE"] = str(int(rng.integers(2)))
    seed = int(rng.integers(1, 1000))
    sp = float(rng.choice([1e-3, 1e-1, 1.0]))
    maxc = 96
    libm = int(rng.integers(4) == 0)                       # the reference's test operator: a quarter of the cases by the formula as written
    os.environ["DAV_HARNESS_LIBM"] = str(libm)
    with fd.CEngine(n=n, max_cols=maxc, gev=kind == "harness_sin") as e:
        e.set_storage(storage)
        if kind.startswith("harness"):
            # the reference's matrix-free test operators (cos: A, sin: B; src/tests/test_utils.f90:72-116) against the oracle's statement
            tab = O.harness_exp_table(n)
            mtx, stx = O.harness_matrices(n)
            e.set_operator_harness(OP_A, tab)
            A = mtx
            if kind == "harness_sin":
                e.set_operator_harness(1, tab)
                A = stx
        elif kind == "host":
            A = rng.standard_normal((n, n)); A = A + A.T
            e.set_dense_host(OP_A, A)
        else:
            A = O.generate_diagonal_dominant(n, sp, seed=seed)
            e.set_dense_generated(OP_A, seed, sp) if kind == "generated" else e.set_operator_hashed(OP_A, seed, sp)
        X = rng.standard_normal((n, maxc))
        e.panel_put(PANEL_V, 0, X)
        msgs = []
        for _ in range(3):
            k = int(rng.integers(1, min(64, maxc) + 1))
            c0 = int(rng.integers(0, maxc - k + 1)); d0 = int(rng.integers(0, maxc - k + 1))
            e.apply(1 if kind == "harness_sin" else OP_A, PANEL_V, c0, k, PANEL_W, d0)
            err = relerr(e.panel_get(PANEL_W, d0, k), A @ X[:, c0:c0 + k])
            if not err < 1e-12 * max(1.0, np.sqrt(n) / 8):
                msgs.append(f"apply k={k} c0={c0} d0={d0} err={err:.2e}")
        # Gram and panel product at random shapes
        Y = rng.standard_normal((n, maxc))
        e.panel_put(PANEL_X, 0, Y)
        for _ in range(3):
            p = int(rng.integers(1, maxc + 1)); q = int(rng.integers(1, 65))
            p0 = int(rng.integers(0, maxc - p + 1)); q0 = int(rng.integers(0, maxc - q + 1))
            G = e.gram(PANEL_V, p0, p, PANEL_X, q0, q)
            err = np.abs(G - X[:, p0:p0 + p].T @ Y[:, q0:q0 + q]).max() / (np.sqrt(n) * 10)
            if not err < 1e-13:
                msgs.append(f"gram p={p} q={q} p0={p0} q0={q0} err={err:.2e}")
            M = rng.standard_normal((p, q))
            d0 = int(rng.integers(0, maxc - q + 1))
            e.panel_transform(PANEL_V, p0, p, M, PANEL_S, d0)
            err = relerr(e.panel_get(PANEL_S, d0, q), X[:, p0:p0 + p] @ M)
            if not err < 1e-13 * max(1.0, p / 4):
                msgs.append(f"panel p={p} q={q} p0={p0} d0={d0} err={err:.2e}")
    bad += bool(msgs)
    print(f"n={n:5d} storage={storage} schedule={sched} {kind + (' libm' if libm and kind.startswith('harness') else ''):16s}: {'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs)}", flush=True)
print(f"{ncases} cases in {time.time() - t0:.0f} s, mismatches: {bad}")
