"""configs[1] (N=20000, lowest=8, DPR; or: N storage lowest max_dim_sub) phase by phase: DAVIDSON_VERBOSE prints the host's wall clock per phase of one solve
(each phase ends in the synchronisation that returns its results, so device time is inside)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch  # noqa
import fortran_davidson_amd as fd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
storage = sys.argv[2] if len(sys.argv) > 2 else "full"
lowest = int(sys.argv[3]) if len(sys.argv) > 3 else 8
max_dim = int(sys.argv[4]) if len(sys.argv) > 4 else None
with fd.DavidsonEngine(n, lowest, max_dim, storage=storage) as eng:
    eng.generate_diagonal_dominant(1, 1e-3, seed=1)
    for _ in range(5):
        eng.solve("DPR", 1000, 1e-8, want_vectors=False)
    os.environ["DAVIDSON_VERBOSE"] = "1"
    for _ in range(4):
        eng.solve("DPR", 1000, 1e-8, want_vectors=False)
