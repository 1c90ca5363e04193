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


def test_scaling_model_prefers_the_dealt_out_symmetric_tiles_and_predicts_six_fold_at_eight():
    one = bench.scaling_model(200000, 16, 1)
    for p in (2, 4, 8):
        m = bench.scaling_model(200000, 16, p)
        assert m["symmetric_overlapped_ms"] <= m["symmetric_ms"] < m["full_ms"]
        assert m["symmetric_ms"] < one["symmetric_ms"] / (0.7 * p)
    eight = bench.scaling_model(200000, 16, 8)
    assert one["symmetric_ms"] / eight["symmetric_ms"] >= 5.7                 # collectives in program order (incl. the start block's reduce-scatter)
    assert one["symmetric_ms"] / eight["symmetric_overlapped_ms"] >= 6.0      # ... on the second stream (the default)


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
