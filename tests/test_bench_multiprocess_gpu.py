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


def run_bench(nproc, extra, env_extra=None):
    env = dict(os.environ, DAVIDSON_TRANSPORT="shm", **(env_extra or {}))
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


_ONE_RANK = {}


@pytest.mark.parametrize("nproc,storage", [(2, "full"), (3, "full"), (2, "symmetric"), (3, "symmetric")])
def test_bench_multi_rank_flow_matches_single_rank(nproc, storage):
    # the timed workload at a reduced order, the same storage on both sides so that the runs are comparable (full
    # row slabs / the lower block triangle dealt out over the ranks: all-gather + reduce-scatter per sweep);
    # the configs[1] / configs[3] / configs[4] legs at small orders
    extra = ["--steps", "2", "--warmup", "1", "--order", "6000", "--storage", storage, "--small-n", "3000", "--gjd-n", "2000",
             "--free-n", "4000", "--harness-n", "0", "--no-cpu-baseline", "--no-dropin", "--all-legs"]
    if storage not in _ONE_RANK:                # the one-rank line of a storage serves both rank counts
        _ONE_RANK[storage] = run_bench(1, extra)
    one = _ONE_RANK[storage]
    many = run_bench(nproc, extra)
    assert many["n_gpus"] == nproc and many["steps"] == 2 and many["scaling"] == "strong"
    assert many["config"]["iters_per_solve"] == one["config"]["iters_per_solve"]
    assert np.abs(np.array(many["eigenvalues"]) - np.array(one["eigenvalues"])).max() < 1e-10
    assert many["roofline"]["achieved"] > 0 and many["roofline"]["hbm_GBps_end_to_end"] > 0
    assert many["roofline"]["hbm_ms_end_to_end"] >= many["roofline"]["hbm_ms_kernel_only"] > 0
    c = many["comm"]
    assert c["sweep_kernel_ms_min_over_ranks"] <= c["sweep_kernel_ms_max_over_ranks"] and c["apply_local_ms_max_over_ranks"] > 0
    assert c["collectives_per_solve"] > 0 and c["model_ms_per_solve"]["symmetric_all_links_ms"] <= c["model_ms_per_solve"]["symmetric_ms"]
    assert many["small"]["phase_ms_per_solve"]["comm_ms"] > 0
    assert many["cpu_baseline"] is None
    for key in ("small", "configs3_gjd", "configs4_free"):
        a, b = one[key], many[key]
        assert "error" not in a and "error" not in b, (a, b)
        assert np.abs(np.array(a["eigenvalues"]) - np.array(b["eigenvalues"])).max() < 1e-8, key
    assert many["small"]["iters_per_solve"] == one["small"]["iters_per_solve"]
    assert many["configs4_free"]["iters"] == one["configs4_free"]["iters"]
    assert np.abs(many["opt_in_policy"]["max_abs_eigenvalue_diff_vs_reference_policy"]) < 1e-8


def test_bench_single_gpu_default_shape_of_the_line():
    """The one-GPU flow with symmetric tiles at a reduced order: every object the contract names is there."""
    line = run_bench(1, ["--steps", "2", "--warmup", "1", "--order", "8000", "--small-n", "3000", "--gjd-n", "3000",
                         "--free-n", "6000", "--harness-n", "6000", "--harness-n2", "9000", "--no-cpu-baseline"])
    assert line["config"]["storage"] == "symmetric" and line["config"]["N"] == 8000
    assert len(line["config"]["workload"]) < 120 and "sparsity" in line["config"]["workload"] and "storage=symmetric" in line["config"]["workload"]
    for key in ("roofline", "apply", "configs3_gjd", "configs4_free", "configs4_free_harness", "small", "dropin", "opt_in_policy", "scaling_model"):
        assert key in line and "error" not in line[key], (key, line.get(key))
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and r["launches"] > 0
    # what the driver's record keeps: the first twenty keys carry both fractions and what they are made of
    head = list(r)[:20]
    assert all(k in head for k in ("frac", "traffic", "hbm_N", "hbm_k", "hbm_frac", "hbm_traffic", "non_kernel_ms_per_solve", "hbm_algorithmic_bytes"))
    assert all(len(v) < 120 for v in r.values() if isinstance(v, str))
    bf = line["configs4_free_harness"]
    assert bf["reference_configuration"]["iters_per_solve"] >= 2 and abs(bf["reference_configuration"]["eigenvalues"][0] - 1.0000992) < 1e-6
    assert 0 < bf["roofline"]["frac"] < 1.2 and bf["large"]["iters"] > 0 and bf["solve_at_configs4_order"]["iters"] > 0
    assert bf["library_call_chain_sweep_ms"] > bf["large"]["ms_per_launch"]            # the one-variable form beats the four library calls
    assert line["scaling_model"]["P8"]["symmetric_all_links_ms"] <= line["scaling_model"]["P8"]["symmetric_ms"]
    # the north-star figure inside `roofline` (what the driver's record keeps), against 8 TB/s and against the rates measured in this run
    assert r["hbm_k"] == 8 and 0 < r["hbm_frac"] <= r["hbm_frac_kernel_only"] < 1
    assert r["hbm_measured_read_GBps"] > 1000 and r["hbm_frac_of_measured_read"] > r["hbm_frac"]
    assert line["hbm_measured"]["read_GBps"] > line["hbm_measured"]["copy_GBps"] * 0.8
    f = line["configs4_free"]["roofline"]
    assert f["launches_of_16_columns"] + f["launches_of_32_columns"] == line["configs4_free"]["launches"] and 0 < f["frac"] < 1.2
    assert line["configs3_gjd"]["sweeps_of_A"] > line["configs3_gjd"]["iters"]
    assert line["dropin"]["iters"] == line["small"]["iters_per_solve"]


def test_bench_multi_rank_flow_with_the_chunked_pipeline_of_wide_blocks():
    """the same launch with the pipeline several GPUs run by default for blocks wider than 32 columns (DAV_SYM_OVERLAP=1 over the
    shared-memory transport: its collectives on the engine's stream, everything else as over RCCL; two-block-row schedule forced
    at this small order): lowest = 16 makes the 64-column expansion; same eigenvalues and iteration count as one rank"""
    extra = ["--steps", "1", "--warmup", "1", "--order", "6000", "--lowest", "16", "--storage", "symmetric", "--headline-only"]
    env = {"DAV_SYM_R": "2", "DAV_SYM_OVERLAP": "1"}
    one = run_bench(1, extra, env)
    for nproc in (2, 3):
        many = run_bench(nproc, extra, env)
        assert many["n_gpus"] == nproc and many["config"]["iters_per_solve"] == one["config"]["iters_per_solve"]
        assert np.abs(np.array(many["eigenvalues"]) - np.array(one["eigenvalues"])).max() < 1e-10
        assert many["comm"]["collectives_per_solve"] > 0 and many["roofline"]["comm_world_size"] == nproc
