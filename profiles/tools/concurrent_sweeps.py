"""Can a generated (compute-bound) sweep hide under a stored (HBM-bound) sweep when both are simply launched at the same time?
(The round-5 verdict's item 4: one pass over X for A and B in the GJD inner solve - the gain it names is hiding B's generated block
rows under A's HBM stalls.)  Two engines on the one GPU, each with its own stream: a stored symmetric matrix (8- and 16-column
sweeps: HBM-bound) and the hashed operator of the same order (generated in the sweep: VALU + MFMA-bound).  Wall time of `reps`
sweeps of each, one after the other, against both at once from two host threads.

    python profiles/tools/concurrent_sweeps.py [N] [k]
"""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch  # noqa: F401
import fortran_davidson_amd as fd
from fortran_davidson_amd.engine_c import OP_A, PANEL_V, PANEL_W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = 6
stored = fd.CEngine(n=n, max_cols=16)
stored.set_storage(1)
stored.set_dense_generated(OP_A, 1, 1e-3)
gen = fd.CEngine(n=n, max_cols=16)
gen.set_storage(1)
gen.set_operator_hashed(OP_A, 2, 1e-3, 1.0)
X = np.random.default_rng(0).standard_normal((n, k))
for e in (stored, gen):
    e.panel_put(PANEL_V, 0, X)
    e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
    e.synchronize()


def run(e):
    for _ in range(reps):
        e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
    e.synchronize()


def timed(fn):
    t0 = time.perf_counter()
    fn()
    return (time.perf_counter() - t0) * 1e3 / reps


t_stored = timed(lambda: run(stored))
t_gen = timed(lambda: run(gen))


def both():
    th = [threading.Thread(target=run, args=(e,)) for e in (stored, gen)]
    [t.start() for t in th]
    [t.join() for t in th]


t_both = timed(both)
print(f"N={n} k={k}: stored sweep {t_stored:.2f} ms, generated sweep {t_gen:.2f} ms, one after the other {t_stored + t_gen:.2f} ms, "
      f"both at once (two streams) {t_both:.2f} ms per pair -> overlap gain {(t_stored + t_gen) / t_both:.3f} x")
stored.close()
gen.close()
