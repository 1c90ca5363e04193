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


def _bench(nproc, extra, env_extra=None, self_launch=False):
    env = dict(os.environ, **(env_extra or {}))
    env.pop("DAVIDSON_TRANSPORT", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if nproc == 1 or self_launch:        # self_launch: bench.py starts torch.distributed.run itself (what `python bench.py --gpus N` does)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
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


def test_two_gpus_with_and_without_overlapped_collectives():
    """Blocks wider than 32 columns (lowest = 16: the 64-column expansion) on two GPUs, collectives in program order on the engine's
    stream (default) and on a second stream under the sweep (DAV_SYM_OVERLAP=1): same eigenvalues and iteration counts as one GPU.
    N = 52000 selects the two-block-row schedule (203 block rows) the overlap applies to.  bench.py starts its own launcher here."""
    if _gpus() < 2:
        pytest.skip("needs two GPUs")
    extra = ["--steps", "1", "--warmup", "1", "--order", "52000", "--lowest", "16", "--storage", "symmetric", "--headline-only"]
    one = _bench(1, extra)
    for overlap in ("0", "1"):
        two = _bench(2, extra, {"DAV_SYM_OVERLAP": overlap}, self_launch=True)
        assert two["n_gpus"] == 2 and two["config"]["iters_per_solve"] == one["config"]["iters_per_solve"]
        assert np.abs(np.array(two["eigenvalues"]) - np.array(one["eigenvalues"])).max() < 1e-10


def test_two_gpus_matrix_free_configs4_shape():
    """BASELINE configs[4] in small: the matrix-free hashed operator, lowest = 8, DPR, generated on two GPUs - as row slabs and as
    dealt-out symmetric tiles (every pair generated once) - against the one-GPU run of the same problem."""
    if _gpus() < 2:
        pytest.skip("needs two GPUs")
    for storage in ("symmetric", "full"):
        extra = ["--steps", "1", "--warmup", "0", "--order", "4000", "--storage", storage, "--free-n", "60000", "--small-n", "0", "--gjd-n", "0",
                 "--restart-sparsity", "0", "--no-dropin", "--no-cpu-baseline", "--all-legs"]
        one, two = _bench(1, extra), _bench(2, extra)
        f1, f2 = one["configs4_free"], two["configs4_free"]
        assert "error" not in f1 and "error" not in f2, (f1, f2)
        assert f2["iters"] == f1["iters"]
        assert np.abs(np.array(f2["eigenvalues"]) - np.array(f1["eigenvalues"])).max() < 1e-10


def test_a_rank_that_dies_ends_the_launch_within_the_bound():
    """One rank of two ends abruptly in the middle of a long run (tests/rank_killer.py: a timer in rank 1): its peer must not wait
    in a collective for ever - the watchdog (DAVIDSON_COLLECTIVE_TIMEOUT) or the launcher ends it, and the launch returns a
    non-zero code well inside the bound."""
    if _gpus() < 2:
        pytest.skip("needs two GPUs")
    import time
    env = dict(os.environ, DAVIDSON_COLLECTIVE_TIMEOUT="20")
    env.pop("DAVIDSON_TRANSPORT", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.time()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29633",
           os.path.join(ROOT, "tests", "rank_killer.py"), "1", "45", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1000000",
           "--warmup", "1", "--order", "6000", "--headline-only"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert res.returncode != 0
    assert time.time() - t0 < 300
