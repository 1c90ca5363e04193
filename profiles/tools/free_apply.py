"""The hashed matrix-free operator (configs[4]) swept on its own: end to end and kernel only per block width
    python profiles/tools/free_apply.py [N] [k,k,...]"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch  # noqa: F401
import fortran_davidson_amd as fd
from fortran_davidson_amd.engine_c import OP_A, PANEL_V

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16, 32, 64]
with fd.CEngine(n=n, max_cols=64) as e:
    e.set_storage(1)
    e.set_operator_hashed(OP_A, 1, 1e-3)
    e.panel_put(PANEL_V, 0, np.random.default_rng(0).standard_normal((n, 64)))
    e.synchronize()
    for k in ks + ks:
        t0 = time.perf_counter()
        ms, kms, nbytes, flops = e.bench_apply2(k, 1)
        e.synchronize()
        print(f"k={k:3d}: {ms:9.1f} ms end to end, {kms:9.1f} ms in the sweep kernels, wall {1e3 * (time.perf_counter() - t0):9.1f} ms", flush=True)
