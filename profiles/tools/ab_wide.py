"""A/B in one process: the wide (one wave per SIMD) symmetric sweep against matvec_sym9_kernel, N=200000, random X."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch  # noqa
import numpy as np
import fortran_davidson_amd as fd
from fortran_davidson_amd.engine_c import OP_A, PANEL_V, PANEL_W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
ks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "16,32,64").split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
with fd.CEngine(n=n, max_cols=64) as e:
    e.set_storage(1)
    e.set_dense_generated(OP_A, 1, 1e-3)
    e.panel_put(PANEL_V, 0, np.random.default_rng(0).standard_normal((n, 64)))
    e.synchronize()
    # same product from both kernels
    for k in ks:
        W = {}
        for wide in ("0", "2"):
            os.environ["DAV_SYM_WIDE"] = wide
            e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
            W[wide] = e.panel_get(PANEL_W, 0, k)
        d = np.abs(W["0"] - W["2"]).max() / np.abs(W["0"]).max()
        print(json.dumps({"check": "wide vs sym9", "n": n, "k": k, "max_rel_diff": float(d)}), flush=True)
        assert d < 1e-13, d
    for r in range(rounds):
        for k in ks:
            for wide in ("0", "2"):
                os.environ["DAV_SYM_WIDE"] = wide
                e.bench_apply2(k, 1)
                ms, kms, nbytes, flops = e.bench_apply2(k, reps)
                print(json.dumps({"round": r, "wide": wide, "n": n, "k": k, "ms": round(ms, 3), "kernel_ms": round(kms, 3),
                                  "GBps_e2e": round(nbytes / ms / 1e6, 1), "TF_kernel": round(flops / kms / 1e9, 2),
                                  "frac_mfma": round(flops / kms / 1e9 / 78.6, 3)}), flush=True)
