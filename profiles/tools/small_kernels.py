"""K2 (Gram / projection) and K3 (panel x small matrix) on their own through the C ABI (dav_gram / dav_panel_transform / the fused
Ritz phase), HIP-event timed by the engine (dav_set_timing level 2: gram_ms / panel_ms), against the bytes each must move:
    python profiles/tools/small_kernels.py [N ...]
Gram p x q: 8 N (p + q) bytes; panel p -> q: 8 N (p + q) bytes; fractions of 8 TB/s."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch  # noqa: F401
import fortran_davidson_amd as fd
from fortran_davidson_amd.engine_c import PANEL_V, PANEL_W, PANEL_S, PANEL_X

orders = [int(a) for a in sys.argv[1:]] or [200000, 20000]
reps = 20
for n in orders:
    rng = np.random.default_rng(1)
    with fd.CEngine(n=n, max_cols=256) as e:
        blk = rng.standard_normal((n, 64))
        for c0 in range(0, 256, 64):
            e.panel_put(PANEL_V, c0, blk)
            e.panel_put(PANEL_W, c0, blk[:, ::-1].copy())
        e.set_timing(2)
        print(f"N={n}")
        for p, q in [(64, 32), (128, 64), (256, 16), (256, 128), (32, 16), (16, 16)]:
            e.gram(PANEL_V, 0, p, PANEL_W, 0, q)
            e.reset_stats()
            for _ in range(reps):
                G = e.gram(PANEL_V, 0, p, PANEL_W, 0, q)
            us = e.stats().gram_ms / reps * 1e3
            nbytes = 8.0 * n * (p + q)
            print(f"  gram  {p:3d} x {q:3d}: {us:8.1f} us  {nbytes / us / 1e6:7.2f} TB/s  {nbytes / us / 1e6 / 8.0:5.3f} of 8 TB/s  "
                  f"{2.0 * n * p * q / us / 1e6:6.2f} TFLOP/s")
        for p, q in [(64, 32), (128, 64), (256, 16), (256, 128), (64, 64), (16, 16)]:
            M = rng.standard_normal((p, q))
            e.panel_transform(PANEL_V, 0, p, M, PANEL_X, 0)
            e.reset_stats()
            for _ in range(reps):
                e.panel_transform(PANEL_V, 0, p, M, PANEL_X, 0)
            e.synchronize()
            us = e.stats().panel_ms / reps * 1e3
            nbytes = 8.0 * n * (p + q)
            print(f"  panel {p:3d} -> {q:3d}: {us:8.1f} us  {nbytes / us / 1e6:7.2f} TB/s  {nbytes / us / 1e6 / 8.0:5.3f} of 8 TB/s  "
                  f"{2.0 * n * p * q / us / 1e6:6.2f} TFLOP/s")
        # in place (the Gram-Schmidt update / restart shape): V[:, 0:q] <- V[:, 0:p] M
        for p, q in [(128, 64), (64, 32)]:
            M = rng.standard_normal((p, q)) / p
            e.reset_stats()
            for _ in range(reps):
                e.panel_transform(PANEL_W, 0, p, M, PANEL_W, 0)
            e.synchronize()
            us = e.stats().panel_ms / reps * 1e3
            print(f"  panel {p:3d} -> {q:3d} in place: {us:8.1f} us")
