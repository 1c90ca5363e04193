import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch  # noqa
import fortran_davidson_amd as fd
from fortran_davidson_amd.engine_c import OP_A, PANEL_V
import numpy as np
n = int(sys.argv[1]); ks = [int(x) for x in sys.argv[2].split(",")]; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
tag = sys.argv[4] if len(sys.argv) > 4 else ""
storage = int(sys.argv[5]) if len(sys.argv) > 5 else 1          # 1 = symmetric tiles, 0 = full rows
with fd.CEngine(n=n, max_cols=64) as e:
    e.set_storage(storage)
    e.set_dense_generated(OP_A, 1, 1e-3)
    e.panel_put(PANEL_V, 0, np.random.default_rng(0).standard_normal((n, 64)))
    e.synchronize()
    for k in ks:
        e.bench_apply2(k, 2)
        ms, kms, nbytes, flops = e.bench_apply2(k, reps)
        print(json.dumps({"tag": tag, "n": n, "k": k, "ms": round(ms, 3), "kernel_ms": round(kms, 3),
                          "GBps_kernel": round(nbytes / kms / 1e6, 1), "frac_mfma": round(flops / kms / 1e9 / 78.6, 3)}), flush=True)
