"""The REAL RCCL path with more than one rank - on the one GPU of the test box.  Every rank is a process on GPU 0 that poses as a
host of its own (NCCL_HOSTID), so RCCL builds a genuine multi-rank communicator and moves the data through its loopback socket
transport (bench.py: DAVIDSON_TRANSPORT=rccl-one-gpu).  Everything above the wire is what a multi-GPU run executes: ncclCommInitRank
with the broadcast id, grouped in-place all-gathers, reduce-scatters, all-reduces, the watchdog's events, the opt-in second-stream
pipeline and the opt-in direct exchanges (ncclSend / ncclRecv) - none of which the loopback / shared-memory test transports or the
1-rank communicator reach.  Launched exactly as the driver launches bench.py for N > 1."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_bench(nproc, extra, env_extra=None, expect_ok=True, timeout=600):
    env = dict(os.environ, DAVIDSON_TRANSPORT="rccl-one-gpu", DAVIDSON_COLLECTIVE_TIMEOUT="120", **(env_extra or {}))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the PRODUCT library (no test transport is involved): not the test build tests/conftest.py points this process at
    env.pop("DAVIDSON_HIP_LIB", None)
    env["LD_LIBRARY_PATH"] = ":".join(p for p in env.get("LD_LIBRARY_PATH", "").split(":") if p and not p.endswith(os.path.join("lib", "test")))
    if nproc == 1:
        env["DAVIDSON_TRANSPORT"] = "rccl"
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    if not expect_ok:
        return res
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


HEAD = ["--steps", "2", "--warmup", "1", "--order", "6000", "--headline-only"]
_ONE_RANK = {}


@pytest.mark.parametrize("nproc,storage", [(2, "symmetric"), (2, "full"), (3, "symmetric"), (4, "symmetric")])
def test_real_rccl_communicator_of_several_ranks_matches_one_rank(nproc, storage):
    if storage not in _ONE_RANK:                # the one-rank line of a storage serves every rank count
        _ONE_RANK[storage] = run_bench(1, HEAD + ["--storage", storage])
    one = _ONE_RANK[storage]
    many = run_bench(nproc, HEAD + ["--storage", storage])
    c = many["comm"]
    assert many["n_gpus"] == nproc and c["transport"] == "rccl-one-gpu" and c["ranks_reported_by_rccl"] == nproc
    assert many["config"]["iters_per_solve"] == one["config"]["iters_per_solve"]
    assert np.abs(np.array(many["eigenvalues"]) - np.array(one["eigenvalues"])).max() < 1e-10
    assert c["collectives_per_solve"] > 0 and c["allgather_ms_per_solve"] > 0 and c["allreduce_ms_per_solve"] > 0
    if storage == "symmetric":
        assert c["reduce_scatter_ms_per_solve"] > 0


def test_the_engine_picks_the_way_of_its_wide_blocks_collectives_itself():
    """lowest = 16 makes the 64-column expansion (two-block-row schedule forced at this small order).  Three ways exist to move its
    collectives: RCCL's all-gather / reduce-scatter in program order on the engine's stream, direct exchanges (grouped ncclSend /
    ncclRecv with every peer and a rank-order sum) and 32-column chunks whose collectives run on a second stream under the sweeps.
    Round 6: the engine decides at its first wide block over a real communicator of several ranks (csrc/engine_apply.hip:
    coll_path_trial) - the block through all three, results compared with the program-order one, times made common with one
    all-reduce, the fastest validated way kept - and reports it (dav_comm_path -> comm.path_*).  Over real 2- and 3-rank communicators:
    the trial runs, validates all three ways (the second stream bitwise), the solve continues on the chosen one to the one-rank
    eigenvalues; a way whose trial result is spoiled (DAV_COLL_TRIAL_CORRUPT: one entry on rank 0) is left out and the run still
    ends correctly; a way forced through the environment (DAV_SYM_OVERLAP / DAV_COLL_DIRECT, or DAV_COLL_SELECT=0) skips the trial."""
    extra = ["--steps", "1", "--warmup", "1", "--order", "6000", "--lowest", "16", "--storage", "symmetric", "--headline-only"]
    one = run_bench(1, extra, {"DAV_SYM_R": "2"})
    names = ("program order", "direct exchange", "second stream")

    def check(many, nproc, what):
        assert many["comm"]["ranks_reported_by_rccl"] == nproc, what
        assert many["config"]["iters_per_solve"] == one["config"]["iters_per_solve"], what
        assert np.abs(np.array(many["eigenvalues"]) - np.array(one["eigenvalues"])).max() < 1e-10, what
        return many["comm"]

    for nproc in (2, 3):
        c = check(run_bench(nproc, extra, {"DAV_SYM_R": "2"}), nproc, "selected by the engine")
        assert c["path_trial_ran"] and c["path_trial_columns"] == 64 and c["path_selected"] in names, c
        assert all(c["path_validated"][w] for w in names), c
        assert all(c["path_trial_ms_max_over_ranks"][w] > 0 for w in names), c
        assert c["collectives_overlapped_with_sweeps"] == (c["path_selected"] == "second stream")
        # a spoiled trial result: that way is left out, the run is unaffected (the kept result is always the program-order one)
        spoiled = 1 if nproc == 3 else 2
        c = check(run_bench(nproc, extra, {"DAV_SYM_R": "2", "DAV_COLL_TRIAL_CORRUPT": str(spoiled)}), nproc, "one way spoiled")
        assert c["path_trial_ran"] and not c["path_validated"][names[spoiled]] and c["path_selected"] != names[spoiled], c
        assert c["path_validated"]["program order"], c
    # forced through the environment: no trial; the second stream gives the bits of program order (same ring reductions, other launch order)
    lams = {}
    for what, env in (("program order", {"DAV_COLL_SELECT": "0"}), ("second stream", {"DAV_SYM_OVERLAP": "1"}), ("direct exchange", {"DAV_COLL_DIRECT": "1"})):
        line = run_bench(2, extra, dict(env, DAV_SYM_R="2"))
        c = check(line, 2, what)
        assert not c["path_trial_ran"] and c["path_selected"] == what, c
        lams[what] = line["eigenvalues"]
    assert lams["second stream"] == lams["program order"]


def test_matrix_free_and_generalized_legs_over_a_real_communicator():
    """The other legs of the bench over a real 2-rank communicator at small orders: configs[1] shape (row slabs), configs[3] shape
    (generalized, GJD, the generated second operator partly resident: its passes agreed on by an all-reduce), configs[4] shape
    (matrix-free, symmetric generation)."""
    extra = ["--steps", "1", "--warmup", "1", "--order", "6000", "--storage", "symmetric", "--small-n", "3000", "--gjd-n", "2000", "--free-n", "4000",
             "--restart-sparsity", "0", "--harness-n", "0", "--no-cpu-baseline", "--no-dropin", "--all-legs"]
    one, two = run_bench(1, extra), run_bench(2, extra)
    for key in ("small", "configs3_gjd", "configs4_free"):
        a, b = one[key], two[key]
        assert "error" not in a and "error" not in b, (key, a, b)
        assert np.abs(np.array(a["eigenvalues"]) - np.array(b["eigenvalues"])).max() < 1e-8, key
    assert two["small"]["iters_per_solve"] == one["small"]["iters_per_solve"] and two["configs4_free"]["iters"] == one["configs4_free"]["iters"]


def test_a_rank_that_dies_inside_a_real_communicator_ends_the_launch_within_the_bound():
    """One rank of two ends abruptly in the middle of a long run (tests/rank_killer.py: a timer in rank 1): its peer must not wait
    in an RCCL collective for ever - the engine's watchdog (DAVIDSON_COLLECTIVE_TIMEOUT) or the launcher ends it, and the launch
    returns a non-zero code well inside the bound."""
    env = dict(os.environ, DAVIDSON_TRANSPORT="rccl-one-gpu", DAVIDSON_COLLECTIVE_TIMEOUT="8")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("DAVIDSON_HIP_LIB", None)
    t0 = time.time()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "rank_killer.py"), "1", "15", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1000000",
           "--warmup", "1", "--order", "6000", "--headline-only"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert res.returncode != 0
    assert time.time() - t0 < 300


def test_configs2_at_full_order_over_a_real_four_rank_communicator():
    """configs[2] at N=200000 on FOUR ranks of a real RCCL communicator (four processes sharing the one GPU: 40 GB of tiles each; the
    box allows six processes on a card, so eight ranks run as threads over the loopback transport instead - tests/test_full_size_gpu.py):
    the iteration count and the eigenvalues of the one-rank run, 10 collectives per solve, 153.6 MB gathered and 204.8 MB
    reduce-scattered per rank and solve through RCCL itself."""
    line = run_bench(4, ["--steps", "1", "--warmup", "1", "--headline-only"], timeout=900)
    c = line["comm"]
    assert line["n_gpus"] == 4 and c["ranks_reported_by_rccl"] == 4 and line["config"]["N"] == 200000 and line["config"]["storage"] == "symmetric"
    assert line["config"]["iters_per_solve"] == 3
    assert np.abs(np.array(line["eigenvalues"]) - np.array([0.9999951655277628, 1.9999960491572697, 2.9999969540010114])).max() < 1e-10
    assert c["collectives_per_solve"] == 10
    assert abs(c["allgather_MB_per_solve"] - 153.6) < 1.0 and abs(c["reduce_scatter_MB_per_solve"] - 204.8) < 1.0
    try:                                                  # kept for profiles/experiments/ (gpurun_out/ travels back from the GPU box)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "r06_rccl_one_gpu_4ranks_n200000.json"), "w") as f:
            json.dump({"note": "bench.py --gpus 4 --headline-only at N=200000 over a REAL 4-rank RCCL communicator whose ranks share one GPU (loopback "
                               "socket transport: the times are of that transport and of four processes time-slicing one device)",
                       "value": line["value"], "iters_per_solve": line["config"]["iters_per_solve"], "eigenvalues": line["eigenvalues"], "comm": c}, f, indent=1)
    except OSError:
        pass


@pytest.mark.parametrize("gev", [False, True])
def test_callers_own_kernel_as_operator_over_a_real_communicator(gev):
    """dav_set_operator_device on two ranks of a real RCCL communicator (tests/rccl_device_operator.py: the banded stencil of
    tests/helpers/user_operator.hip, n = 2999 - it reads across the slab boundary, so the block the callback is handed must be the
    gathered one): the eigenvalues and the iteration count of the dense solve of the same matrix by numpy / the oracle."""
    from oracle import davidson_oracle as O
    n, lowest = 2999, 4
    a = np.diag(1.0 + np.arange(n, dtype=np.float64))
    b = np.eye(n)
    for off, w in ((1, 0.3), (2, 0.15)):
        a += w * (np.eye(n, k=off) + np.eye(n, k=-off))
    for off, w in ((1, 0.05), (2, 0.025)):
        b += w * (np.eye(n, k=off) + np.eye(n, k=-off))
    with np.errstate(invalid="ignore"):
        lam_o, _, it_o = O.generalized_eigensolver_dense(a, lowest, "DPR", 200, 1e-8, None, b if gev else None)
    env = dict(os.environ, DAVIDSON_COLLECTIVE_TIMEOUT="120")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # (the helper library travels in lib/test/; the engine is whichever library this process's environment points at)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "rccl_device_operator.py"), str(n), str(lowest), "1" if gev else "0"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert line["ranks"] == 2 and line["iters"] == it_o and line["collectives"] > 0
    assert np.abs(np.array(line["eigenvalues"]) - lam_o).max() < 1e-9
