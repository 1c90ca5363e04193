"""The real multi-GPU data path: one process per GPU, RCCL all-gather / all-reduce / reduce-scatter over xGMI, launched
exactly as the driver launches bench.py for N > 1.  Needs at least two GPUs: skipped on the one-GPU test box (where the
same flow runs over the shared-memory transport, tests/test_bench_multiprocess_gpu.py, and the RCCL calls through a
1-rank communicator, tests/test_solver_gpu.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus():
    import torch
    return torch.cuda.device_count()


def _bench(nproc, extra):
    env = dict(os.environ)
    env.pop("DAVIDSON_TRANSPORT", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    # a rank that fails must not leave its peers waiting in a collective for ever: the whole launch is bounded
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.parametrize("storage", ["symmetric", "full"])
def test_two_gpus_over_rccl_match_one_gpu(storage):
    if _gpus() < 2:
        pytest.skip("needs two GPUs")
    extra = ["--steps", "2", "--warmup", "1", "--order", "6000", "--storage", storage, "--headline-only"]
    one, two = _bench(1, extra), _bench(2, extra)
    assert two["n_gpus"] == 2
    assert two["config"]["iters_per_solve"] == one["config"]["iters_per_solve"]
    assert np.abs(np.array(two["eigenvalues"]) - np.array(one["eigenvalues"])).max() < 1e-10
