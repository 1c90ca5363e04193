"""The reference's matrix-free test operator (src/tests/test_utils.f90:72-116) generated in the symmetric sweep, on its own:
    python profiles/tools/harness_apply.py [N] [k1,k2,...]
HIP-event time per launch, entries evaluated per second, against the measured rate of the same arithmetic on registers
(dav_bench_harness_rate)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch  # noqa: F401
import fortran_davidson_amd as fd
from fortran_davidson_amd.engine_c import OP_A, PANEL_V, PANEL_W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8,16,32").split(",")]
i = np.arange(1, n + 1, dtype=np.float32)
tab = np.exp(i / np.float32(n), dtype=np.float32).astype(np.float64)
with fd.CEngine(n=n, max_cols=max(max(ks), 16)) as e:
    rate = e.bench_harness_rate(3000)
    print(f"arithmetic on registers: {rate:.4g} entries/s")
    e.set_storage(1)
    e.set_operator_harness(OP_A, tab)
    e.panel_put(PANEL_V, 0, np.random.default_rng(0).standard_normal((n, max(ks))))
    for k in ks:
        e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
        e.synchronize(); e.reset_stats()
        for _ in range(3):
            e.apply(OP_A, PANEL_V, 0, k, PANEL_W, 0)
        e.synchronize()
        st = e.stats()
        ms = st.apply_kernel_ms / max(st.apply_launches, 1)
        ent = 0.5 * n * (n + 1.0) * (max(1, (k + 15) // 16) if k > 16 else 1)
        print(f"N={n} k={k:2d}: {st.apply_launches // 3} launch(es), {ms:8.2f} ms per launch, {ent / (st.apply_launches // 3) / (ms * 1e-3):.4g} entries/s "
              f"= {ent / (st.apply_launches // 3) / (ms * 1e-3) / rate:.3f} of the register rate")
