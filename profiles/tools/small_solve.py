import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch  # noqa
import numpy as np
import fortran_davidson_amd as fd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for storage in ("full", "symmetric"):
    for R in (("0",) if storage == "full" else ("1", "2")):
        if R != "0":
            os.environ["DAV_SYM_R"] = R
        with fd.DavidsonEngine(n, 8, None, storage=storage) as eng:
            eng.generate_diagonal_dominant(1, 1e-3, seed=1)
            for _ in range(5):
                eng.solve("DPR", 1000, 1e-8, want_vectors=False)
            eng.c.synchronize(); eng.c.set_timing(2); eng.c.reset_stats()
            t0 = time.perf_counter()
            its = 0
            for _ in range(50):
                lam, _, it = eng.solve("DPR", 1000, 1e-8, want_vectors=False)
                its += it
            eng.c.synchronize()
            dt = (time.perf_counter() - t0) / 50
            st = eng.c.stats()
            print(json.dumps({"n": n, "storage": storage, "R": R, "ms_per_solve": round(dt * 1e3, 4), "iters": its // 50,
                              "apply_ms": round(st.apply_ms / 50, 4), "apply_kernel_ms": round(st.apply_kernel_ms / 50, 4),
                              "gram_ms": round(st.gram_ms / 50, 4), "panel_ms": round(st.panel_ms / 50, 4), "lam0": float(lam[0])}), flush=True)
