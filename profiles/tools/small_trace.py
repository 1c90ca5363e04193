"""configs[1] (N=20000, lowest=8, DPR, full storage): 20 warm solves for a kernel trace (rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch  # noqa
import fortran_davidson_amd as fd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
storage = sys.argv[2] if len(sys.argv) > 2 else "full"
with fd.DavidsonEngine(n, 8, None, storage=storage) as eng:
    eng.generate_diagonal_dominant(1, 1e-3, seed=1)
    for _ in range(20):
        eng.solve("DPR", 1000, 1e-8, want_vectors=False)
