"""CPU: the parts of bench.py that need no GPU - the launch plumbing, the CPU-share detection of the baseline leg, the scaling
model behind `--storage auto`, and the refusal to start a launcher from under a profiler."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_cpu_share_is_what_the_job_may_use():
    share, quota = bench.cpu_share()
    assert 1 <= share <= (os.cpu_count() or 1)
    assert quota is None or quota > 0


def test_scaling_model_from_the_one_gpu_rehearsal_of_eight_ranks():
    """bench.scaling_model on the rehearsal log (P ranks taking turns on one GPU, profiles/experiments/r06_ranks_rehearsal_n200000.jsonl):
    per-rank times are MEASURED, the link rate is the assumption.  With the collectives in program order the model clears six-fold
    at 8 GPUs where the exchanges use the direct links of the mesh, and stays under it if every collective were a ring on ONE link."""
    assert bench.rehearsal_inputs().keys() >= {1, 2, 4, 8}
    one = bench.scaling_model(200000, 16, 1)
    assert one["speedup_symmetric"] == 1.0
    last = one["symmetric_ms"]
    for p in (2, 4, 8):
        m = bench.scaling_model(200000, 16, p)
        assert m["inputs"]["per_rank_times_from"] == "rehearsal"
        assert max(m["symmetric_all_links_ms"], m["symmetric_overlapped_ms"]) <= m["symmetric_ms"] < last
        assert m["symmetric_ms"] < one["symmetric_ms"] / (0.65 * p)
        last = m["symmetric_ms"]
    eight = bench.scaling_model(200000, 16, 8)
    assert eight["speedup_symmetric_all_links"] >= 6.0
    assert 5.0 <= eight["speedup_symmetric"] < eight["speedup_symmetric_overlapped"] < eight["speedup_symmetric_all_links"]
    assert eight["inputs"]["collectives_per_solve"] <= 10
    # inputs measured in the run itself take precedence (a faster box scales the per-rank times)
    fast = bench.scaling_model(200000, 16, 8, {"apply_ms": 0.9 * bench.rehearsal_inputs()[1]["apply_local_ms"], "host_ms": 1.0, "ms_per_solve": 113.0})
    assert fast["inputs"]["per_rank_sweeps_end_to_end_ms"] < eight["inputs"]["per_rank_sweeps_end_to_end_ms"]


def test_no_self_launch_from_under_a_profiler():
    """`python bench.py --gpus 2` starts torch.distributed.run as a child - unless a profiler's preloaded library has already
    initialised the GPU in this process: then it stops with a message instead of making the forbidden launcher hop."""
    env = dict(os.environ, ROCPROFILER_REGISTER_FORCE_LOAD="1", ROCP_TOOL_LIBRARIES="/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    for k in ("RANK", "WORLD_SIZE"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--control-plane-only"], capture_output=True, text=True,
                         timeout=120, env=env)
    assert res.returncode != 0 and "will not start its own launcher under a profiler" in (res.stderr + res.stdout)


def test_control_plane_of_the_multi_rank_launch():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--control-plane-only", "--order", "1000"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 0, (res.stdout + res.stderr)[-2000:]
    assert '"control_plane": "ok"' in res.stdout
