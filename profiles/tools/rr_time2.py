"""DSYEVD (all pairs) against DSYEVR on the leading quarter, alternating, many repetitions (sequential MKL on this host)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(__file__))
os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("rr", os.path.join(os.path.dirname(__file__), "rr_time.py"))
# reuse the ctypes wrappers without running the module's timing loops
src = open(spec.origin).read().split("rng = np.random.default_rng(0)")[0]
ns = {}
exec(src, ns)
rng = np.random.default_rng(1)
cases = [(64, 16), (96, 24), (128, 32), (128, 16), (160, 40), (256, 64), (256, 32), (256, 16), (400, 100), (400, 50), (512, 32), (800, 200), (800, 100), (800, 32)]
for n, k in cases:
    Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
    A = Q @ np.diag(np.arange(1, n + 1) + 1e-3 * rng.standard_normal(n)) @ Q.T; A = (A + A.T) / 2
    t = {"dsyevd": [], "dsyevr_subset": [], "dsyev": []}
    for rep in range(100 if n <= 256 else 10):
        for name, f in (("dsyevd", lambda: ns["dsyevd"](A)), ("dsyevr_subset", lambda: ns["dsyevr"](A, k)), ("dsyev", lambda: ns["dsyev"](A))):
            t0 = time.perf_counter(); f(); t[name].append(time.perf_counter() - t0)
    print(f"n={n:4d} k={k:3d}: " + "  ".join(f"{name} median {np.median(v) * 1e6:8.1f} us (min {np.min(v) * 1e6:7.1f})" for name, v in t.items()), flush=True)
