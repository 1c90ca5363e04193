"""One-off evidence run (not part of the test suite): the engine's choice of the way of its wide blocks' collectives over REAL RCCL
communicators of 4 and 5 ranks on the one GPU (every rank a process posing as a host of its own, as tests/test_rccl_one_gpu.py does
for 2 and 3 ranks; the box allows six GPU processes).  Prints the `comm` section of each bench line.
usage: python profiles/tools/selector_ranks.py [ranks ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_rccl_one_gpu import run_bench  # noqa: E402

extra = ["--steps", "1", "--warmup", "1", "--order", "6000", "--lowest", "16", "--storage", "symmetric", "--headline-only"]
one = run_bench(1, extra, {"DAV_SYM_R": "2"})
for nproc in [int(a) for a in sys.argv[1:]] or [4, 5]:
    line = run_bench(nproc, extra, {"DAV_SYM_R": "2"})
    c = line["comm"]
    diff = max(abs(a - b) for a, b in zip(line["eigenvalues"], one["eigenvalues"]))
    print(json.dumps({"ranks": nproc, "ranks_reported_by_rccl": c["ranks_reported_by_rccl"], "iters": line["config"]["iters_per_solve"],
                      "iters_one_rank": one["config"]["iters_per_solve"], "max_abs_eigenvalue_diff_vs_one_rank": diff,
                      "path_trial_ran": c["path_trial_ran"], "path_selected": c["path_selected"], "path_validated": c["path_validated"],
                      "path_trial_ms_max_over_ranks": c["path_trial_ms_max_over_ranks"], "collectives_per_solve": c["collectives_per_solve"]}), flush=True)
