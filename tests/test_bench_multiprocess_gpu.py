"""bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process per rank),
on a single-GPU box: the ranks share GPU 0 and exchange through the shared-memory test transport
(DAVIDSON_TRANSPORT=shm), everything else - rendezvous, engine per process with its row slab, all-gather of
the new basis block, all-reduce of the small results, barriers, max-over-ranks timing, the JSON line - is
the code the multi-GPU run executes.  The RCCL calls themselves are covered by the 1-rank communicator test."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_bench(nproc, extra):
    env = dict(os.environ, DAVIDSON_TRANSPORT="shm")
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("nproc", [2, 3])
def test_bench_multi_rank_flow_matches_single_rank(nproc):
    extra = ["--steps", "2", "--warmup", "1", "--order", "6000", "--large-n", "5000", "--no-cpu-baseline"]
    one = run_bench(1, extra)
    many = run_bench(nproc, extra)
    assert many["n_gpus"] == nproc and many["steps"] == 2 and many["scaling"] == "strong"
    assert many["config"]["iters_per_solve"] == one["config"]["iters_per_solve"]
    assert np.abs(np.array(many["eigenvalues"]) - np.array(one["eigenvalues"])).max() < 1e-10
    assert many["roofline"]["achieved"] > 0 and many["phase_ms_per_step"]["comm_ms"] > 0
    assert many["cpu_baseline"] is None
    big1, bigp = one["large"], many["large"]
    assert "error" not in bigp, bigp
    assert bigp["iters_per_solve"] == big1["iters_per_solve"]
    assert np.abs(np.array(bigp["eigenvalues"]) - np.array(big1["eigenvalues"])).max() < 1e-10
    assert np.abs(many["opt_in_policy"]["max_abs_eigenvalue_diff_vs_reference_policy"]) < 1e-8
