"""Accuracy of the error-free slicing that profiles/ubench/slice_i8.hip prices (exploratory, round 5): a 256 x 256 off-diagonal tile of
generate_diagonal_dominant (entries U[0,1) * 1e-3) times 32 columns, by slice products as an int8 matrix core would form them - every
slice product exact in integers, the 8 partial sums of equal s + t folded in fp64 - against the exact product (Python integers).
    python profiles/tools/slice_accuracy.py            (host only: no GPU)
Error metric of the parity tests: |W - A X| relative to sum_j |a_ij| |x_j| (tests/test_full_size_gpu.py: ROW_TOL 1e-12)."""
import os
import sys
from fractions import Fraction
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from oracle import davidson_oracle as O

BITS = 7


def slices(v, e_scale, nslices):
    """v = 2**e_scale * sum_s q_s 2**(-BITS s), q_s integers with |q_s| < 2**BITS (sign carried by every slice); exact up to
    2**(e_scale - BITS nslices)"""
    t = np.abs(v) / 2.0 ** e_scale                      # in [0, 1): exact (power of two)
    sign = np.sign(v).astype(np.int64)
    out = []
    for _ in range(nslices):
        t = t * 2.0 ** BITS
        q = np.floor(t)
        t = t - q                                        # exact
        out.append(q.astype(np.int64) * sign)
    return out


def run(A, X, nslices, what):
    n, k = X.shape
    eA = int(np.ceil(np.log2(np.abs(A).max() * (1 + 2.0 ** -50))))                  # one exponent per TILE
    eX = np.ceil(np.log2(np.abs(X).max(axis=0) * (1 + 2.0 ** -50))).astype(int)     # one per column of the block
    As = slices(A, eA, nslices)
    Xs = [np.stack([slices(X[:, j], int(eX[j]), nslices)[s] for j in range(k)], axis=1) for s in range(nslices)]
    W = np.zeros((A.shape[0], k))
    npairs = 0
    for u in range(2 * nslices, 1, -1):                  # smallest terms first; s, t 1-based
        if u > nslices + 1:
            continue
        P = np.zeros((A.shape[0], k), dtype=np.int64)
        for s in range(1, nslices + 1):
            t = u - s
            if 1 <= t <= nslices:
                P += As[s - 1] @ Xs[t - 1]               # exact: |entries| < 256 * 2**14 * 8
                npairs += 1
        W += P.astype(np.float64) * 2.0 ** (-BITS * u)
    W = W * 2.0 ** eA * (2.0 ** eX)[None, :]
    # exact reference on a sample of rows (Python rationals)
    rows = np.arange(0, A.shape[0], 16)
    worst = 0.0
    for i in rows:
        for j in range(0, k, 5):
            exact = sum(Fraction(float(A[i, l])) * Fraction(float(X[l, j])) for l in range(n))
            scale = float(np.abs(A[i]) @ np.abs(X[:, j]))
            worst = max(worst, abs(float(Fraction(float(W[i, j])) - exact)) / scale)
    ref = A @ X
    fp64 = max(abs(float(Fraction(float(ref[i, j])) - sum(Fraction(float(A[i, l])) * Fraction(float(X[l, j])) for l in range(n)))) /
               float(np.abs(A[i]) @ np.abs(X[:, j])) for i in rows[:4] for j in range(0, k, 8))
    print(f"{what}: {nslices} slices of {BITS} bits, {npairs} slice products: max error / sum|a||x| = {worst:.2e}   (plain fp64 dot products: {fp64:.2e})")


n = 4096
A_rows = O.generate_diagonal_dominant(n, 1e-3, seed=1, rows=np.arange(512, 768))      # rows 512..767: tile (2, J)
tile = A_rows[:, 1024:1280]                                                           # an off-diagonal tile
rng = np.random.default_rng(0)
X = rng.standard_normal((256, 32))
Xb = X.copy(); Xb[:3] *= 1e6; Xb[10:200] *= 1e-9                                      # basis-vector-like: a few large components, most tiny
for ns in (7, 8):
    run(tile, X, ns, "random X")
    run(tile, Xb, ns, "basis-like X")
