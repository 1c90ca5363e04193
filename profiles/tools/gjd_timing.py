"""configs[3] (N=200000 generalized, GJD, lowest=8; B hashed, partly resident) timed at the engine's timing levels 0 / 1 / 2
(level 2 records an event pair around every phase) and, with DAVIDSON_VERBOSE, by host phase:
    python profiles/tools/gjd_timing.py [N]"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch  # noqa: F401
import fortran_davidson_amd as fd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
with fd.DavidsonEngine(n, 8, 80, gev=True, storage="symmetric") as g:
    g.generate_diagonal_dominant(1, 1e-3, seed=1)
    g.set_hashed_operator(2, 1e-3, 1.0, seed=2)
    g.solve("GJD", 1000, 1e-8, want_vectors=False)
    print("resident fraction of B", g.c.resident_fraction(1), flush=True)
    for level in (0, 2, 0, 1, 2):
        g.c.set_timing(level)
        g.c.synchronize(); g.c.reset_stats()
        t0 = time.perf_counter()
        lam, _, it = g.solve("GJD", 1000, 1e-8, want_vectors=False)
        g.c.synchronize()
        dt = time.perf_counter() - t0
        st = g.c.stats()
        print(f"timing level {level}: {dt * 1e3:8.1f} ms, iters {it}, applies {st.applies}, apply_ms {st.apply_ms:.1f}, panel_ms {st.panel_ms:.1f}, "
              f"gram_ms {st.gram_ms:.1f}", flush=True)
    os.environ["DAVIDSON_VERBOSE"] = "1"
    g.c.set_timing(0)
    g.solve("GJD", 1000, 1e-8, want_vectors=False)
